"""Per-block timeline of conv3x3_c64.hip (profiling build: python hallucidet_amd/build.py --trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from hallucidet_amd import _abi
_abi.LIB_PATH = os.environ.get("HD_TRACE_LIB") or os.path.join(os.path.dirname(_abi.LIB_PATH), "libhallucidet_hip_trace.so")
lib = _abi.load()
lib.hd_conv_trace_buffer.restype = C.c_int
lib.hd_conv_trace_buffer.argtypes = [C.c_void_p]
from hallucidet_amd import ops
dev = "cuda"
med = lambda a: float(np.median(a))

for name, N, H, W, stats in [("layer1 fwd", 8, 128, 160, True), ("layer1 dgrad", 8, 128, 160, False)]:
    x = torch.randn(N, H, W, 64, device=dev, dtype=torch.float16)
    w = (torch.randn(64, 576, device=dev) * 0.05).half()
    for tid in (0, 192):
        os.environ["HD_TRACE_TID"] = str(tid)
        lib.hd_conv_trace_buffer(None)
        for _ in range(3):
            ops.conv2d(x, w, 3, 3, pad=1, want_stats=stats)
        buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
        lib.hd_conv_trace_buffer(buf.data_ptr())
        ops.conv2d(x, w, 3, 3, pad=1, want_stats=stats)
        torch.cuda.synchronize()
        lib.hd_conv_trace_buffer(None)
        t = buf.cpu().numpy().reshape(-1, 16)
        t = t[t[:, 0] != 0]
        if len(t) == 0:
            print(name, "no stamps"); continue
        w0, c0, c2, c3, c4, c5, w1, hw, tb, tk, tw, te, nt, c13 = [t[:, i] for i in range(14)]
        span = (w1.max() - w0.min()) / 100.0
        print("%-14s tid %3d: %d blocks, span %.1f us | dma issue %5.0f | weights+patch land %5.0f | weights->regs %5.0f | per tile (%.1f tiles): barrier %5.0f  K loop %5.0f  vmcnt %5.0f  epilogue %5.0f | tile loop total %6.0f | stats tail %5.0f | life %6.0f clk" %
              (name, tid, len(t), span, med(c2 - c0), med(c3 - c2), med(c4 - c3), med(nt), med(tb / nt), med(tk / nt), med(tw / nt), med(te / nt), med(c5 - c4), med(c13 - c5), med(c13 - c0)))
