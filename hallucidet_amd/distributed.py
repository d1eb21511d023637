"""Data parallelism for the hallucination network: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI, gradients of the 24.4 M U-Net parameters only (the detector is frozen and replicated).

All parameters and gradients live in ONE flat fp32 arena (UnetRunner.flatten_parameters), so the exchange is a handful of
large all-reduces over contiguous slices (xGMI is point-to-point, per-link bound: few big messages beat many small ones).
OVERLAP: the backward pass completes the arena from its end (head + decoder, then layer4 ... layer1 + stem), and the runner
calls `bucket_ready(lo, hi)` at its exchange boundaries -- by default TWO (UnetRunner.bucket_ranges, unet.py): decoder + layer4 +
layer3 (92 MB, 94 % of the arena, final after ~60 % of the backward pass: its all-reduce runs under layer2 / layer1 / the stem)
and the rest (5.6 MB, the only exposed part); they are the boundaries at which the deferred weight-gradient grids run, so the
data-parallel step keeps the single-GPU launch schedule.  HD_EXCHANGE_BUCKETS=5 restores one bucket per backward segment
(rounds 2-4).  `finish()` waits right before the optimizer step; the division by the world size rides in the fused Adam's
inverse scale (`exchange_and_step`: no extra pass over the 97.75 MB arena).  Without hooks `start()` issues the whole arena after
backward (n_buckets slices).  BatchNorm statistics stay per rank (the reference has no SyncBatchNorm).  The same code runs on the
gloo backend for CPU tests.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    # HD_FORCE_DIST=1: run the collectives even at world size 1 (exercises the RCCL path on a single-GPU box)
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("HD_FORCE_DIST") == "1")


def broadcast_parameters(flat_params, buffers=()):
    """DDP start-up semantics: rank 0's parameters and BatchNorm buffers everywhere."""
    if not is_dist():
        return
    dist.broadcast(flat_params, src=0)
    for b in buffers:
        dist.broadcast(b, src=0)


class GradientAverager:
    def __init__(self, n_buckets=4):
        self.n_buckets = n_buckets
        self._work = []
        self._flat = None
        self._covered = []
        self.issued = []          # [(lo, hi)] of the current step, in issue order (tests / diagnostics)
        self.timing = False       # bench.py: HIP events around every bucket's wait -> exposed (un-overlapped) communication time
        self._events = []         # [[(e0, e1, nbytes)] per step]

    def begin(self, flat_grads):
        """Start of a step's backward pass: buckets will arrive through bucket_ready()."""
        self._flat, self._covered, self._work, self.issued = flat_grads, [], [], []

    def bucket_ready(self, lo, hi):
        """Gradient slice [lo, hi) is final: launch its SUM all-reduce now (async; ordered after the kernels already enqueued on
        the current stream, concurrent with everything enqueued afterwards)."""
        if self._flat is None or hi <= lo:
            return
        self._covered.append((lo, hi))
        self.issued.append((lo, hi))
        if is_dist():
            self._work.append(dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def start(self, flat_grads):
        """Launch SUM all-reduces (async) for whatever part of the arena the bucket hooks have not covered yet (all of it when
        no hook ran: n_buckets equal slices)."""
        if self._flat is not flat_grads:
            self.begin(flat_grads)
        n = flat_grads.numel()
        done = sorted(self._covered)
        gaps, pos = [], 0
        for lo, hi in done:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < n:
            gaps.append((pos, n))
        for lo, hi in gaps:
            step = (hi - lo + self.n_buckets - 1) // self.n_buckets
            step = max(1024, (step + 1023) // 1024 * 1024)
            for o in range(lo, hi, step):
                self.bucket_ready(o, min(hi, o + step))

    def finish(self, flat_grads, defer_mean=False):
        """Wait for the all-reduces; -> the factor that turns the arena's content into the MEAN gradient (DDP semantics).
        defer_mean=False: the arena is multiplied here and 1.0 is returned.  defer_mean=True: the arena keeps the SUM over ranks
        and 1 / world is returned -- the caller hands it to the fused optimizer as its inverse scale (`g * inv_scale` is the first
        thing hd_adam_step computes per element, so the separate read + write of the whole arena is not needed; same products)."""
        work, self._work, self._flat = self._work, [], None
        if not is_dist():
            return 1.0
        if self.timing and flat_grads.is_cuda:
            # w.wait() makes the CURRENT stream wait for the collective (the host does not block): the time between two events
            # recorded on that stream around it is the part of the bucket's all-reduce that the backward pass did not hide
            ev = []
            for w, (lo, hi) in zip(work, self.issued):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                w.wait()
                e1.record()
                ev.append((e0, e1, (hi - lo) * 4))
            self._events.append(ev)
        else:
            for w in work:
                w.wait()
        inv = 1.0 / dist.get_world_size()
        if defer_mean:
            return inv
        flat_grads.mul_(inv)
        return 1.0

    def exposed_wait_ms(self):
        """Mean exposed wait per bucket (issue order) over the steps timed so far: [{'bytes', 'ms'}]; call after a synchronize."""
        if not self._events:
            return []
        n = min(len(e) for e in self._events)
        out = []
        for k in range(n):
            ms = [e[k][0].elapsed_time(e[k][1]) for e in self._events]
            out.append({"bytes": self._events[0][k][2], "ms": round(sum(ms) / len(ms), 4)})
        return out


def exchange_and_step(averager, flat_grads, scaler, optimizer):
    """What follows backward() in a data-parallel training step (EncoderDecoderLit.fit_step): cover the slices no bucket hook
    reported, wait for every all-reduce, then the (replicated) overflow check / mean (x 1 / world, inside the fused Adam) / clip /
    Adam step and the GradScaler update.  Every rank sees the same summed gradient -- an inf on one rank is an inf on all -- so the
    skip decision, the loss scale and the parameters stay identical across ranks without any further communication.  After the
    call `flat_grads` holds the SUM over ranks (the mean was never written)."""
    averager.start(flat_grads)
    inv = averager.finish(flat_grads, defer_mean=True)
    scaler.step(optimizer, inv_scale=inv)
    scaler.update()
