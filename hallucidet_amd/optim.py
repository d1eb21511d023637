"""Optimizer and loss scaling for the hallucination network.

`FusedAdam` = torch.optim.Adam semantics (reference: config.py:204-245 builds Adam(lr); train_hallucidet.py:431-435)
with Lightning's `gradient_clip_val=0.5, gradient_clip_algorithm="value"` (train_hallucidet.py:498-499) folded in, run
as ONE kernel over the flat fp32 parameter arena (hd_adam_step).  `LossScaler` mirrors torch.cuda.amp.GradScaler's
policy (init 2**16, growth x2 every 2000 clean steps, backoff x0.5), which Lightning installs for `--precision 16`.
"""
import torch

from . import ops


class ParamArena:
    """Flat fp32 parameter / gradient arenas over an arbitrary parameter list (the detector's trainable parameters in
    train_detector.py; the U-Net has its own inside UnetRunner).  Every `p.data` / `p.grad` becomes a view, so one fused
    optimizer launch updates everything and a data-parallel exchange is a few large all-reduces over contiguous memory."""

    def __init__(self, params):
        self._params = list(params)
        self._flat = None
        self.grad_scale = 1.0
        self.flatten_parameters()

    def flatten_parameters(self):
        ps = self._params
        if self._flat is not None and all(p.data_ptr() == self._flat.data_ptr() + o * 4 and p.grad is not None
                                          and p.grad.data_ptr() == self._gflat.data_ptr() + o * 4
                                          for p, o in ((ps[0], self._offsets[0]), (ps[-1], self._offsets[-1]))):
            return
        dev = ps[0].device
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = gflat[o:o + p.numel()].view(p.shape)
        self._flat, self._gflat, self._offsets = flat, gflat, offs

    @property
    def flat_params(self):
        self.flatten_parameters()
        return self._flat

    @property
    def flat_grads(self):
        self.flatten_parameters()
        return self._gflat


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, unet, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0):
        """`unet`: a module with a `.runner` that owns flat arenas (Unet), or a ParamArena."""
        self.runner = unet if isinstance(unet, ParamArena) else unet.runner
        self.runner.flatten_parameters()
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_value=clip_value)
        super().__init__(list(self.runner._params), defaults)
        flat = self.runner.flat_params
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0
        self.skipped_steps = 0
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=flat.device)
        # the skip decision is made on the device (hd_adam_step returns early when found_inf is set); the host learns of it
        # through this pinned word, copied behind the optimizer kernel and read at the next step's first host wait
        self._inf_host = torch.zeros(1, dtype=torch.float32).pin_memory() if flat.is_cuda else torch.zeros(1)
        self._inf_event = torch.cuda.Event() if flat.is_cuda else None
        self._inf_pending = False
        self._stash = False          # result of a flag resolved inside step() that no LossScaler.resolve() has seen yet

    def zero_grad(self, set_to_none=False):
        # gradients are overwritten (not accumulated) by every backward pass; nothing to do, and the views must stay
        return None

    @torch.no_grad()
    def step(self, closure=None, inv_scale=1.0, check_inf=False):
        r = self.runner
        r.flatten_parameters()
        g = self.param_groups[0]
        if self._inf_pending:               # the previous step's skip flag must be accounted for before it is overwritten;
            self._stash = self.resolve_found_inf()   # a LossScaler.resolve() that comes later still learns of it
        if check_inf:
            self.found_inf.zero_()
            ops.check_finite(r.flat_grads, self.found_inf)
        self.step_count += 1
        ops.adam_step(r.flat_params, r.flat_grads, self.exp_avg, self.exp_avg_sq, lr=g["lr"], beta1=g["betas"][0],
                      beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"], clip_value=g["clip_value"],
                      inv_scale=inv_scale, step=self.step_count, found_inf=self.found_inf if check_inf else None)
        if check_inf:
            self._inf_host.copy_(self.found_inf, non_blocking=True)
            if self._inf_event is not None:
                self._inf_event.record()
            self._inf_pending = True

    def resolve_found_inf(self):
        """-> True if the LAST `step(check_inf=True)` was skipped on the device (non-finite gradient).  Waits only for that
        step's own kernels (an event behind the pinned copy), which have long finished by the time the next step asks.  A
        skipped step does not count: `step_count` (Adam's bias correction) is rolled back, as torch.optim.Adam never sees it."""
        if not self._inf_pending:
            bad, self._stash = self._stash, False
            return bad
        if self._inf_event is not None:
            self._inf_event.synchronize()
        self._inf_pending = False
        bad = bool(self._inf_host.item() != 0.0)
        if bad:
            self.step_count -= 1
            self.skipped_steps += 1
        return bad


class LossScaler:
    def __init__(self, unet, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.runner = unet if isinstance(unet, ParamArena) else unet.runner
        self.scale_value = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._good = 0
        self._optimizer = None
        self._update_due = False

    def scale(self, loss):
        self.resolve()                      # GradScaler order: the scale of step t+1 reflects step t's overflow check
        self.runner.grad_scale = self.scale_value
        return loss * self.scale_value

    def scale_tensor(self, like):
        """The current scale as a persistent 0-dim device tensor (rewritten only when the value changes)."""
        t = self.__dict__.get("_scale_t")
        if t is None or t.device != like.device or t.dtype != like.dtype:
            t = self._scale_t = torch.empty((), dtype=like.dtype, device=like.device)
            self._scale_t_value = None
        if self._scale_t_value != self.scale_value:
            t.fill_(self.scale_value)
            self._scale_t_value = self.scale_value
        return t

    def backward(self, loss):
        """`scale(loss).backward()` without its three launches (loss * scale, the ones_like seed, the seed * scale of MulBackward): the
        backward pass is seeded with the scale itself, d(scale * loss) / d loss."""
        self.resolve()
        self.runner.grad_scale = self.scale_value
        loss.backward(gradient=self.scale_tensor(loss))

    def step(self, optimizer, inv_scale=1.0):
        """Parameter gradients were already divided by the scale in-kernel; only inf/nan detection remains.  `inv_scale`: a factor
        the optimizer applies to every gradient element first (1 / world size of a data-parallel SUM exchange)."""
        self.resolve()
        self._optimizer = optimizer
        if inv_scale == 1.0:
            optimizer.step(check_inf=self.enabled)
        else:
            optimizer.step(check_inf=self.enabled, inv_scale=inv_scale)
        return optimizer.found_inf

    def update(self, found_inf_host=None):
        """GradScaler.update().  With `found_inf_host` (python bool: the caller already knows) the policy is applied now;
        otherwise it is applied by `resolve()` -- called from the next `scale()` / `step()` -- which reads the device flag of
        the step just issued WITHOUT a new host synchronisation point in this step."""
        if not self.enabled:
            return
        if found_inf_host is None:
            self._update_due = True
            return
        if self._optimizer is not None:
            self._optimizer.resolve_found_inf()
        self._update_due = False
        self._apply(bool(found_inf_host))

    def resolve(self):
        """Apply the pending update() (if any) from the optimizer's device flag; -> whether that step was skipped."""
        if self._optimizer is None:
            return False
        bad = self._optimizer.resolve_found_inf()
        if self._update_due:
            self._update_due = False
            self._apply(bad)
        return bad

    def _apply(self, bad):
        if bad:
            self.scale_value *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale_value *= self.growth_factor
                self._good = 0
