"""Host logic of the Lightning stand-in (hallucidet_amd/trainer.py) and of the loss scaler (hallucidet_amd/optim.py) that needs no
GPU: eval-mode validation / test loops, patience semantics of Lightning 1.5.10's EarlyStopping, GradScaler policy
driven by the optimizer's device flag."""
import torch
import torch.nn as nn

from hallucidet_amd.optim import LossScaler
from hallucidet_amd.trainer import Trainer


class _Lit:
    def __init__(self):
        self.encoder_decoder = nn.Sequential(nn.Conv2d(1, 2, 1), nn.BatchNorm2d(2))
        self.detector = nn.BatchNorm2d(2)
        self.seen_modes = []
        self.optimizer = object()
        self.val_maps = []

    def fit_step(self, batch, i):
        self.encoder_decoder.train()
        self.encoder_decoder(batch[0])
        return torch.tensor(1.0)

    def validation_step(self, batch, i):
        self.seen_modes.append((self.encoder_decoder.training, self.detector.training))
        with torch.no_grad():
            self.detector(self.encoder_decoder(batch[0]))
        return torch.tensor(2.0)

    test_step = validation_step

    def on_validation_epoch_end(self):
        return {"map": torch.tensor(self.val_maps.pop(0) if self.val_maps else 0.5)}

    on_test_epoch_end = on_validation_epoch_end

    def save_checkpoint(self, *a, **k):
        return None


class _DM:
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.b = [(tuple(torch.randn(1, 8, 8, generator=g) for _ in range(4)), tuple({} for _ in range(4))) for _ in range(3)]

    def train_dataloader(self):
        return self.b

    val_dataloader = test_dataloader = train_dataloader


def test_validation_and_test_run_in_eval_mode_and_leave_bn_statistics_alone():
    lit, dm = _Lit(), _DM()
    tr = Trainer(max_epochs=1, device="cpu", monitor="val_map", log=lambda *a: None)
    lit.encoder_decoder.train()
    lit.detector.train()
    before = {k: v.clone() for k, v in list(lit.encoder_decoder.state_dict().items()) + list(lit.detector.state_dict().items())}
    m = tr.validate(lit, dm.val_dataloader())
    tr.test(lit, dm)
    assert lit.seen_modes and all(mode == (False, False) for mode in lit.seen_modes)
    after = dict(list(lit.encoder_decoder.state_dict().items()) + list(lit.detector.state_dict().items()))
    assert all(torch.equal(before[k], after[k]) for k in before), "running statistics moved during validation"
    assert lit.encoder_decoder.training and lit.detector.training                  # previous modes restored
    assert abs(m["val_loss"] - 2.0) < 1e-12 and m["val_map"] == 0.5
    tr.fit(lit, dm)                                                                  # training does update the statistics
    assert not torch.equal(before["1.running_mean"], lit.encoder_decoder.state_dict()["1.running_mean"])


def test_early_stopping_patience_counts_like_lightning():
    """EarlyStopping(patience=2): best at epoch 0, no improvement at epochs 1 and 2 -> stop after epoch 2 (wait_count >= patience)."""
    lit, dm = _Lit(), _DM()
    lit.val_maps = [0.9, 0.1, 0.1, 0.1, 0.1, 0.1]
    tr = Trainer(max_epochs=6, device="cpu", monitor="val_map", early_stopping=("val_map", "max", 2), log=lambda *a: None)
    assert len(tr.fit(lit, dm)) == 3


class _FakeOpt:
    def __init__(self):
        self.step_count, self.flags, self.calls = 0, [], 0
        self.found_inf = None
        self._pending = None

    def step(self, check_inf=False):
        self.step_count += 1
        self._pending = self.flags.pop(0)

    def resolve_found_inf(self):
        bad, self._pending = bool(self._pending), None
        if bad:
            self.step_count -= 1
        return bad


class _R:
    grad_scale = 1.0

    def __init__(self):
        self.runner = self


def test_loss_scaler_follows_gradscaler_policy_from_the_optimizer_flag():
    sc, opt = LossScaler(_R(), init_scale=1024.0, growth_interval=3), _FakeOpt()
    opt.flags = [False, True, False, False, False, True]
    scales = []
    for _ in range(6):
        scales.append(float(sc.scale(torch.tensor(1.0))))       # scale() of step t+1 applies step t's verdict
        sc.step(opt)
        sc.update()
    sc.resolve()
    # step 1 overflowed -> halve, clean-step counter restarts; three clean steps -> double; the last overflow halves again
    assert scales == [1024.0, 1024.0, 512.0, 512.0, 512.0, 1024.0] and sc.scale_value == 512.0
    assert opt.step_count == 4                                   # skipped steps do not advance Adam's bias correction
