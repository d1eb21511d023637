"""Per-block timeline of the 64-deep implicit-GEMM conv kernel on the hot path's shapes (profiling build only).

Build first (cross-compiles here):  python hallucidet_amd/build.py --trace
Each block's wave 0 stamps: wall clock at entry/exit (100 MHz), shader clock at entry, after the prologue DMA issue, after the
first tile landed, after the K loop, after the epilogue, and the CU it ran on.  Prints, per shape: launch span, how many blocks
each CU ran, and the median cycles per phase -- which part of a short-K launch is fixed cost and which is the K loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from hallucidet_amd import _abi

_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libhallucidet_hip_trace.so")
lib = _abi.load()
lib.hd_conv_trace_buffer.restype = C.c_int
lib.hd_conv_trace_buffer.argtypes = [C.c_void_p]
from hallucidet_amd import ops

dev = "cuda"
SHAPES = [   # name, N, H, W, Cin, Cout, KH, want_stats, override(bm,bn,bk,deep) or None
    ("layer1 64->64 @128x160", 8, 128, 160, 64, 64, 3, True, None),
    ("layer2 128->128 @64x80", 8, 64, 80, 128, 128, 3, True, None),
    ("layer3 256->256 @32x40", 8, 32, 40, 256, 256, 3, True, None),
    ("layer4 512->512 @16x20", 8, 16, 20, 512, 512, 3, True, None),
    ("det 256->256 @75x75 x24", 24, 75, 75, 256, 256, 3, False, None),
    ("det 256->256 @38x38 x24", 24, 38, 38, 256, 256, 3, False, None),
    ("det 64->64 3x3 @75x75 x24", 24, 75, 75, 64, 64, 3, False, None),
    ("det 256->64 1x1 @75x75 x24", 24, 75, 75, 256, 64, 1, False, None),
    ("det 64->256 1x1 @75x75 x24", 24, 75, 75, 64, 256, 1, False, None),
    ("det 256->256 3x3 @19x19 x24", 24, 19, 19, 256, 256, 3, False, None),
    ("det 1024->256 1x1 @19x19 x24", 24, 19, 19, 1024, 256, 1, False, None),
    ("layer1 as 128x64 deep", 8, 128, 160, 64, 64, 3, True, (128, 64, 64, 1)),
    ("layer1 as 64x64", 8, 128, 160, 64, 64, 3, True, (64, 64, 64, 0)),
]
CLK_GHZ = 2.4


def med(a):
    return float(np.median(a))


for name, N, H, W, Cin, Cout, KH, stats, ov in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev, dtype=torch.float16)
    w = (torch.randn(Cout, KH * KH * Cin, device=dev) * 0.05).half()
    if ov is None:
        lib.hd_conv_tune_override(-1, -1, -1, -1)
    else:
        lib.hd_conv_tune_override(*ov)
    lib.hd_conv_trace_buffer(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.conv2d(x, w, KH, KH, pad=KH // 2, want_stats=stats)
    e0.record()
    for _ in range(10):
        ops.conv2d(x, w, KH, KH, pad=KH // 2, want_stats=stats)
    e1.record(); e1.synchronize()
    t_us = e0.elapsed_time(e1) * 100
    nblk_max = 1 << 16
    buf = torch.zeros(nblk_max * 16, dtype=torch.int64, device=dev)
    lib.hd_conv_trace_buffer(buf.data_ptr())
    ops.conv2d(x, w, KH, KH, pad=KH // 2, want_stats=stats)
    torch.cuda.synchronize()
    lib.hd_conv_trace_buffer(None)
    t = buf.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 0] != 0]
    nb = len(t)
    if nb == 0:
        print("%-28s no stamps (not a 64-deep launch)" % name)
        continue
    wall0, c0, c_pro, c_land, c_loop, c_end, wall1, hw, c_tr, c_rows = [t[:, i] for i in range(10)]
    span_us = (wall1.max() - wall0.min()) / 100.0
    cu = (hw & 0xF00) >> 8
    se = (hw >> 13) & 0x7
    xcc = (hw >> 32) & 0xF
    sh = (hw >> 12) & 1
    cuid = xcc * 1000 + se * 32 + sh * 16 + cu
    uniq, cnt = np.unique(cuid, return_counts=True)
    flops = 2.0 * N * H * W * Cout * KH * KH * Cin
    nk = KH * KH * Cin // 64
    life = (c_end - c0)
    print("%-28s %6.1f us/launch (%4.0f TF/s)  traced span %6.1f us  blocks %5d on %3d CUs (%d..%d per CU)  nk=%d" %
          (name, t_us, flops / t_us / 1e6, span_us, nb, len(uniq), cnt.min(), cnt.max(), nk))
    print("    median cycles: setup+prologue issue %5.0f | first tile lands %5.0f | K loop %6.0f (%4.0f per step) | epilogue %5.0f | block life %6.0f = %.2f us" %
          (med(c_pro - c0), med(c_land - c_pro), med(c_loop - c_land), med(c_loop - c_land) / nk, med(c_end - c_loop), med(life), med(life) / CLK_GHZ / 1e3))
    if t.shape[1] >= 14 and (t[:, 10] != 0).any():
        s10, s11, s12, s13 = [t[:, i] for i in (10, 11, 12, 13)]
        print("    setup split: entry -> tile mapping / descriptors %5.0f | A rows %5.0f | B rows + acc init %5.0f | first pixel state %5.0f | prologue DMA issue %5.0f" %
              (med(s10 - c0), med(s11 - s10), med(s12 - s11), med(s13 - s12), med(c_pro - s13)))
    print("    epilogue split: acc->LDS transpose + barrier %5.0f | row loop (LDS read, math, stores) %5.0f | BN partial sums %5.0f" %
          (med(c_tr - c_loop), med(c_rows - c_tr), med(c_end - c_rows)))
    if os.environ.get("TRACE_CU"):
        for c in uniq[:2]:
            sel = np.where(cuid == c)[0]
            order = sel[np.argsort(wall0[sel])]
            print("    CU %d: " % c + "  ".join("[%.1f-%.1f]" % ((wall0[i] - wall0.min()) / 100.0, (wall1[i] - wall0.min()) / 100.0) for i in order))
    start_us = (wall0 - wall0.min()) / 100.0
    print("    block start times: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us;   sum of block lives / (CUs x span) = %.2f blocks in flight per CU" %
          (np.percentile(start_us, 10), np.percentile(start_us, 50), np.percentile(start_us, 90), start_us.max(),
           life.sum() / CLK_GHZ / 1e3 / (len(uniq) * span_us)))
