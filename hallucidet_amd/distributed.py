"""Data parallelism for the hallucination network: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI, gradients of the 24.4 M U-Net parameters only (the detector is frozen and replicated).

All parameters and gradients live in ONE flat fp32 arena (UnetRunner.flatten_parameters), so the exchange is a handful of
large all-reduces over contiguous slices (default 4 buckets of ~24 MB: xGMI is point-to-point, per-link bound, so few big
messages beat many small ones) issued asynchronously and waited on right before the optimizer step.  BatchNorm statistics
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

    def start(self, flat_grads):
        """Launch bucketed SUM all-reduces (async)."""
        self._work = []
        if not is_dist():
            return
        n = flat_grads.numel()
        step = (n + self.n_buckets - 1) // self.n_buckets
        step = (step + 1023) // 1024 * 1024
        for o in range(0, n, step):
            self._work.append(dist.all_reduce(flat_grads[o:min(n, o + step)], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self, flat_grads):
        """Wait and divide by the world size (mean gradient, as DDP)."""
        if not is_dist():
            return
        for w in self._work:
            w.wait()
        self._work = []
        flat_grads.mul_(1.0 / dist.get_world_size())
