"""Where the HOST spends a training step, single-GPU vs the data-parallel code path at world size 1 (HD_FORCE_DIST=1): perf_counter stamps
around the sections of EncoderDecoderLit._fit_step, averaged over steady-state steps, next to the step's wall time.  The GPU is never
waited for inside a section, so a section's time is host work (or a blocking call).
    python tools/probe_dist_host.py            # single
    HD_FORCE_DIST=1 python tools/probe_dist_host.py"""
import os
import sys
import time
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if os.environ.get("HD_FORCE_DIST") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
from hallucidet_amd import synthetic, distributed
from hallucidet_amd import train_hallucidet as th

if os.environ.get("HD_PROBE_AR") in ("none", "tiny", "handshake", "handshake_kernel"):
    # what the collective CALL costs at world size 1: "none" = the hooks run, no collective is launched; "tiny" = every bucket's collective
    # moves four bytes (same stream hand-shakes, no payload)
    import torch.distributed as dist

    class _Done:
        def wait(self):
            return True
    _real = dist.all_reduce
    _tiny = None

    def _fake(t, op=None, async_op=False, **k):
        global _tiny
        if os.environ["HD_PROBE_AR"] == "none":
            return _Done()
        if os.environ["HD_PROBE_AR"].startswith("handshake"):
            # what ProcessGroupNCCL does around a collective, without one: the side stream waits for the compute stream, (a trivial
            # kernel,) the compute stream later waits for the side stream
            global _side
            if "_side" not in globals():
                _side = torch.cuda.Stream(priority=-1)
            cur = torch.cuda.current_stream()
            _side.wait_stream(cur)
            if os.environ["HD_PROBE_AR"] == "handshake_kernel":
                with torch.cuda.stream(_side):
                    if _tiny is None:
                        _tiny = torch.zeros(1, device=t.device)
                    _tiny.add_(1.0)

            class _W:
                def wait(self_):
                    torch.cuda.current_stream().wait_stream(_side)
                    return True
            return _W()
        if _tiny is None:
            _tiny = torch.zeros(1, device=t.device)
        return _real(_tiny, op=op, async_op=async_op)
    dist.all_reduce = _fake
    distributed.dist.all_reduce = _fake
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
acc = collections.OrderedDict()
on = [False]


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            if on[0]:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return w


r = lit.encoder_decoder.runner
lit.training_step = timed("training_step (U-Net fwd + detector graphs enqueued)", lit.training_step)
r.run_backward = timed("  runner.run_backward (graph replays + hooks)", r.run_backward)
lit.averager.bucket_ready = timed("    averager.bucket_ready (all_reduce launch)", lit.averager.bucket_ready)
lit.averager.start = timed("  averager.start", lit.averager.start)
lit.averager.finish = timed("  averager.finish (waits)", lit.averager.finish)
lit.scaler.step = timed("  scaler.step (check_finite + Adam)", lit.scaler.step)
lit.scaler.scale = timed("scaler.scale (resolves the previous step's flag)", lit.scaler.scale)
th.exchange_and_step = timed("exchange_and_step", th.exchange_and_step)
orig_backward = torch.Tensor.backward
torch.Tensor.backward = timed("loss.backward() (autograd: detector + U-Net backward enqueued)", orig_backward)
for _ in range(6):
    lit.fit_step(batch)
torch.cuda.synchronize()
on[0] = True
K = 30
t0 = time.perf_counter()
for _ in range(K):
    lit.fit_step(batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("mode: %s   step wall %.3f ms (host returned after %.3f ms per step)" % ("HD_FORCE_DIST=1" if distributed.is_dist() else "single", t_all / K * 1e3, t_host / K * 1e3))
for k, v in acc.items():
    print("  %-70s %8.1f us per step" % (k, v / K * 1e6))
