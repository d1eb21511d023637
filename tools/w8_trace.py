"""Per-block timeline of the 8-wave conv family (profiling build: python hallucidet_amd/build.py --trace).
Wave 0 of every block stamps entry / setup done / first stage landed / K loop done / exit, and accumulates, over its K steps, the
clocks spent in: MEM phase work, the barrier after it, MFMA phase (incl. its vmcnt wait), the barrier after it."""
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
SHAPES = [   # name, N, H, W, Cin, Cout, KH, want_stats
    ("det 256->256 @75x75 x24", 24, 75, 75, 256, 256, 3, False),
    ("layer3 256->256 @32x40", 8, 32, 40, 256, 256, 3, True),
    ("layer2 128->128 @64x80", 8, 64, 80, 128, 128, 3, True),
    ("layer1 64->64 @128x160", 8, 128, 160, 64, 64, 3, True),
    ("det 256->256 @19x19 x24", 24, 19, 19, 256, 256, 3, False),
    ("det 256->256 @19x19 x8", 8, 19, 19, 256, 256, 3, False),
    ("det 512->512 @10x10 x24", 24, 10, 10, 512, 512, 3, False),
][int(os.environ.get("SHAPE_LO", 0)):]
CFGS = [int(v) for v in os.environ.get("CFGS", "10,11,12").split(",")]     # tiles of the patch-staged family (conv3x3_w8.hip)
TILES = [(256, 128), (128, 128), (256, 64), (128, 64), (128, 256), (64, 128), (64, 256), None, None, None, (256, 128), (128, 128), (256, 64), (128, 64),
         None, (128, 128), (256, 64), (128, 64), (160, 64), (320, 64), (96, 64)]        # 15..17: the step-split main loop (TS) of 11..13; 18 / 19 / 20: conv3x3_m160.hip
med = lambda a: float(np.median(a))
for name, N, H, W, Cin, Cout, KH, stats in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev, dtype=torch.float16)
    w = (torch.randn(Cout, KH * KH * Cin, device=dev) * 0.05).half()
    for cfg in CFGS:
        if TILES[cfg][1] // 2 >= max(Cout, 64) and TILES[cfg][1] > 64:
            continue
        lib.hd_conv_tune_w8(cfg, 1)
        lib.hd_conv_trace_buffer(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.conv2d(x, w, KH, KH, pad=KH // 2, want_stats=stats)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, w, KH, KH, pad=KH // 2, want_stats=stats)
        e1.record(); e1.synchronize()
        t_us = e0.elapsed_time(e1) * 100
        buf = torch.zeros((1 << 16) * 16, dtype=torch.int64, device=dev)
        lib.hd_conv_trace_buffer(buf.data_ptr())
        ops.conv2d(x, w, KH, KH, pad=KH // 2, want_stats=stats)
        torch.cuda.synchronize()
        lib.hd_conv_trace_buffer(None)
        t = buf.cpu().numpy().reshape(-1, 16)
        t = t[t[:, 0] != 0]
        if len(t) == 0:
            print(name, cfg, "no stamps")
            continue
        wall0, c0, c_set, c_land, c_loop, c_end, wall1, hw, a_mem, a_b1, a_mfma, a_b2, e_wr, e_rows, e_x, e_y = [t[:, i] for i in range(16)]
        nk = KH * KH * Cin // 64
        span_us = (wall1.max() - wall0.min()) / 100.0
        clk = med((c_end - c0) / np.maximum(wall1 - wall0, 1)) * 100 / 1e3   # GHz
        flops = 2.0 * N * H * W * Cout * KH * KH * Cin
        print("%-26s cfg %d %dx%d  %6.1f us/launch (%4.0f TF/s) span %6.1f us blocks %4d nk=%d clk %.2f GHz" % (
            name, cfg, TILES[cfg][0], TILES[cfg][1], t_us, flops / t_us / 1e6, span_us, len(t), nk, clk))
        print("    median clocks: setup %5.0f | first stage %5.0f | K loop %6.0f = %4.0f/step [mem %4.0f, bar %4.0f, mfma+vmcnt %4.0f, bar %4.0f] | epilogue %5.0f | life %6.0f" % (
            med(c_set - c0), med(c_land - c_set), med(c_loop - c_land), med(c_loop - c_land) / nk, med(a_mem) / nk, med(a_b1) / nk,
            med(a_mfma) / nk, med(a_b2) / nk, med(c_end - c_loop), med(c_end - c0)))
        if e_wr.any():
            print("    epilogue split: acc -> LDS + loads issued + barrier %5.0f | row loop %5.0f | BN sums %5.0f" % (
                med(e_wr - c_loop), med(e_rows - e_wr), med(c_end - e_rows)))
            if e_x.any():
                print("    BN sums split: wait for all waves' rows %5.0f | partials -> LDS + barrier %5.0f | column sums + store %5.0f" % (
                    med(e_x - e_rows), med(e_y - e_x), med(c_end - e_y)))
lib.hd_conv_tune_w8(-1, 0)
