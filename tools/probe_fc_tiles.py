import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, _abi
lib = _abi.load()
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
dev = "cuda"
cases = [("fc6 dgrad", (4096, 1, 1, 1024), (12544, 1024), 1, {}),
         ("fc6 fwd 8192", (8192, 7, 7, 256), (1024, 12544), 7, {"act": 1}),
         ("fc6 fwd 4096", (4096, 7, 7, 256), (1024, 12544), 7, {"act": 1}),
         ("fc7 fwd", (8192, 1, 1, 1024), (1024, 1024), 1, {"act": 1})]
for name, xs, ws, k, kw in cases:
    x = torch.randn(*xs, device=dev, dtype=torch.float16) * 0.1
    w = torch.randn(*ws, device=dev, dtype=torch.float16) * 0.02
    fl = 2.0 * xs[0] * ws[0] * ws[1]
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    base = t(lambda: ops.conv2d(x, w, k, k, **kw))
    out = ["%s: shipped %.1f us (%.0f TF/s)" % (name, base, fl / base / 1e6)]
    for bm in (64, 128):
        for bn in (64, 128):
            for deep in (0, 1):
                lib.hd_conv_tune_override(bm, bn, 64, deep)
                tt = t(lambda: ops.conv2d(x, w, k, k, **kw))
                out.append("%dx%d%s %.1f" % (bm, bn, "d" if deep else "", tt))
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    print(" | ".join(out))

# (the 8-wave im2col family that was probed here -- 166-270 us on the fc6 data gradient against 165 for the 128x64 4-wave tile -- was removed in round 3)
