"""Minimal stand-in for `pl.Trainer` (pytorch_lightning 1.5.10 is not installable offline; SURVEY 8b "Script surface"): drives
the reference's hook names on the MI355X modules -- training_step (through fit_step = backward / all-reduce / optimizer),
validation_step, on_validation_epoch_end, test_step, on_test_epoch_end, ReduceLROnPlateau on `val_loss`, best / last
checkpoints in Lightning's layout -- over `hallucidet_amd.dataloader` data modules staged to HBM by DevicePrefetcher.
What the reference's three scripts configure on the Trainer and what happens here:
  max_epochs, limit_train_batches                       -> same meaning
  EarlyStopping(monitor, mode, patience)                 -> `early_stopping=(monitor, mode, patience)`
  ModelCheckpoint(monitor=..., filename="best")          -> `<dirpath>/best.ckpt` on improvement of the monitor
  gradient_clip_val / precision                          -> owned by the module (FusedAdam clip, LossScaler)
"""
import math
import os

import torch

from .dataloader import DevicePrefetcher


class Trainer:
    def __init__(self, max_epochs=10, limit_train_batches=1.0, dirpath=None, monitor="val_map", mode="max", early_stopping=None,
                 device="cuda", log_every=50, log=print):
        self.max_epochs, self.limit_train_batches = max_epochs, limit_train_batches
        self.dirpath, self.monitor, self.mode = dirpath, monitor, mode
        self.early_stopping = early_stopping
        self.device, self.log_every, self.log = device, log_every, log
        self.current_epoch, self.global_step = 0, 0
        self.best = None
        self.history = []

    def _better(self, a, b):
        return b is None or (a > b if self.mode == "max" else a < b)

    @staticmethod
    def _flat(metrics):
        out = {}
        for k, v in metrics.items():
            if isinstance(v, dict):
                for k2, v2 in v.items():
                    if torch.is_tensor(v2) and v2.numel() == 1:
                        out["%s/%s" % (k, k2)] = float(v2)
            elif torch.is_tensor(v) and v.numel() == 1:
                out[k] = float(v)
        return out

    @staticmethod
    def _modules(model):
        return [m for m in (getattr(model, "encoder_decoder", None), getattr(model, "detector", None)) if m is not None]

    def _set_eval(self, model):
        """Lightning runs the validation / test loops under `model.eval()` + no_grad and restores the previous modes: BatchNorm
        uses (and does not update) its running statistics, the detector runs its eval-mode RPN (1000/1000) and transform."""
        mods = self._modules(model)
        was = [m.training for m in mods]
        for m in mods:
            m.eval()
        return list(zip(mods, was))

    @staticmethod
    def _restore(state):
        for m, was in state:
            m.train(was)

    @staticmethod
    def _mean_over_ranks(value, count):
        """Every rank must take the same LR-schedule / early-stop / best-checkpoint decisions: average the monitored scalars."""
        from .distributed import is_dist
        if not is_dist():
            return value / max(count, 1)
        import torch.distributed as dist
        t = torch.tensor([float(value), float(count)], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        return float(t[0]) / max(float(t[1]), 1.0)

    @staticmethod
    def _nanmean_over_ranks(metrics):
        """All monitored scalars in ONE unconditional all-reduce: per key a (sum, count) pair, NaN / 'no data' (-1 from the mAP
        accumulators) entries contribute (0, 0).  Every rank issues the same collective whatever its local values are -- a
        per-key reduce guarded by a local NaN test would pair up differently on ranks whose values differ."""
        from .distributed import is_dist
        keys = sorted(metrics)
        pairs = []
        for k in keys:
            v = metrics[k]
            s, c = (float(v[0]), float(v[1])) if isinstance(v, tuple) else (float(v), 1.0)
            if math.isnan(s) or (k.startswith("map") and s == -1.0):
                s, c = 0.0, 0.0
            pairs += [s, c]
        if is_dist():
            import torch.distributed as dist
            t = torch.tensor(pairs, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t)
            pairs = t.tolist()
        out = {}
        for i, k in enumerate(keys):
            s, c = pairs[2 * i], pairs[2 * i + 1]
            if c > 0:
                out[k] = s / c
            else:
                v = metrics[k]
                out[k] = float("nan") if isinstance(v, tuple) else float(v)      # nobody had data: keep the local sentinel
        return out

    @staticmethod
    def _flush_deferred_checks(model):
        det = getattr(model, "detector", None)
        if det is not None:
            from .utils.eval_forward_fasterrcnn import flush_degenerate
            flush_degenerate(det, block=True)

    def validate(self, model, loader):
        self._flush_deferred_checks(model)          # the last training batch's degenerate-box flag is read here, not during validation
        state = self._set_eval(model)
        try:
            tot, n = 0.0, 0
            for i, batch in enumerate(DevicePrefetcher(loader, self.device)):
                r = model.validation_step(batch, i)
                tot += float(r[0] if isinstance(r, tuple) else r)
                n += 1
            m = self._flat(model.on_validation_epoch_end())
            self._flush_deferred_checks(model)
        finally:
            self._restore(state)
        m.update(self._nanmean_over_ranks(dict(m, val_loss=(tot, n))))
        # Lightning logs `val_map` = the hallucinated stream's mAP (train_hallucidet.py:357) / the detector's mAP
        m["val_map"] = m.get("map_hall/map", m.get("map", float("nan")))
        return m

    def fit(self, model, datamodule):
        if getattr(model, "optimizer", None) is None:
            model.prepare()
        train, val = datamodule.train_dataloader(), datamodule.val_dataloader()
        n_train = max(1, int(math.floor(len(train) * self.limit_train_batches))) if self.limit_train_batches <= 1.0 else int(self.limit_train_batches)
        bad = 0
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            run, seen = 0.0, 0
            for i, batch in enumerate(DevicePrefetcher(train, self.device)):
                if i >= n_train:
                    break
                loss = model.fit_step(batch, i)
                self.global_step += 1
                if (i + 1) % self.log_every == 0 or i + 1 == n_train:
                    run, seen = float(loss), i + 1           # one host sync per log line, not per step
                    self.log("epoch %d step %d/%d loss %.5f" % (epoch, seen, n_train, run))
            m = self.validate(model, val)
            m.update(epoch=epoch, train_loss=run)
            if hasattr(model, "lr_scheduler_step"):
                m["lr"] = model.lr_scheduler_step(m["val_loss"])
            self.history.append(m)
            self.log("epoch %d " % epoch + " ".join("%s=%.5g" % (k, v) for k, v in sorted(m.items()) if isinstance(v, float)))
            cur = m.get(self.monitor)
            if cur is not None and not math.isnan(cur) and self._better(cur, self.best):
                self.best, bad = cur, 0
                if self.dirpath:
                    os.makedirs(self.dirpath, exist_ok=True)
                    model.save_checkpoint(os.path.join(self.dirpath, "best.ckpt"), epoch=epoch, global_step=self.global_step)
            else:
                bad += 1
            # Lightning 1.5.10 EarlyStopping: stop when wait_count >= patience
            if self.early_stopping and bad >= self.early_stopping[2]:
                self.log("early stop: %s did not improve for %d epochs" % (self.monitor, bad))
                break
        return self.history

    def save_checkpoint(self, model, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        return model.save_checkpoint(path, epoch=self.current_epoch, global_step=self.global_step)

    def test(self, model, datamodule):
        state = self._set_eval(model)
        try:
            for i, batch in enumerate(DevicePrefetcher(datamodule.test_dataloader(), self.device)):
                model.test_step(batch, i)
            self._flush_deferred_checks(model)
            return model.on_test_epoch_end()
        finally:
            self._restore(state)
