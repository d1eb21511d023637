"""Data parallelism for the hallucination network: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI, gradients of the 24.4 M U-Net parameters only (the detector is frozen and replicated).

All parameters and gradients live in ONE flat fp32 arena (UnetRunner.flatten_parameters), so the exchange is a handful of
large all-reduces over contiguous slices (xGMI is point-to-point, per-link bound: few big messages beat many small ones).
OVERLAP: the backward pass completes the arena from its end (head + decoder, then layer4 ... layer1 + stem), and the runner
calls `bucket_ready(lo, hi)` at each of those five boundaries -- the all-reduce of a bucket is issued while the kernels of the
following encoder stages still run (the 52 MB layer4 bucket is ready after ~35 % of the backward) and `finish()` waits right
before the optimizer step.  Without hooks `start()` issues the whole arena after backward (4 buckets).  BatchNorm statistics
stay per rank (the reference has no SyncBatchNorm).  The same code runs on the gloo backend for CPU tests.
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

    def finish(self, flat_grads):
        """Wait and divide by the world size (mean gradient, as DDP)."""
        work, self._work, self._flat = self._work, [], None
        if not is_dist():
            return
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
        flat_grads.mul_(1.0 / dist.get_world_size())

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
    reported, wait for every all-reduce, take the mean, then the (replicated) unscale / overflow check / clip / Adam step and the
    GradScaler update.  Every rank sees the same averaged gradient -- an inf on one rank is an inf on all -- so the skip decision,
    the loss scale and the parameters stay identical across ranks without any further communication."""
    averager.start(flat_grads)
    averager.finish(flat_grads)
    scaler.step(optimizer)
    scaler.update()
