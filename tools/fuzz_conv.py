"""Randomised parity sweep of the convolution entry points (hd_conv2d forward / data gradient, hd_wgrad + hd_wgrad_reduce, hd_conv2d_wgrad,
hd_conv2d_multi) against ATen's fp32 convolution ON THE GPU (an implementation this repository did not write), over shapes drawn from the
whole domain the dispatcher accepts: odd extents, ragged tiles, every kernel size / stride on the path, channel counts on both sides of
every tile boundary, the nearest-2x + concat gather, bias / residual / ReLU / sigmoid epilogues.  The fixed cases of
tests/test_kernels_gpu.py pin the variants one by one; this looks for the combination nobody wrote down.

    python tools/fuzz_conv.py [--cases 300] [--seed 0] [--max-seconds 600]

Prints one line per failing case (with the seed that reproduces it) and a summary; exit code 1 if anything failed."""
import argparse
import math
import os
import random
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CH_IN = [8, 16, 24, 32, 40, 64, 72, 96, 128, 192, 256, 320, 512]
CH_OUT = [3, 8, 16, 24, 32, 40, 64, 72, 128, 136, 192, 256, 512]


def rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, device="cuda") * scale).half()


def err_of(got, want, rtol, atol):
    got, want = got.float(), want.float()
    e = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = e > tol
    return int(bad.sum()), float(e.max()) if e.numel() else 0.0


def nchw(t):
    return t.float().permute(0, 3, 1, 2)


def gathered_input(x, x2, up1):
    a = nchw(x)
    if up1:
        a = a.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    if x2 is not None:
        a = torch.cat([a, nchw(x2)], dim=1)
    return a


def draw_case(r):
    K, stride, pad = r.choice([(3, 1, 1)] * 6 + [(3, 2, 1)] * 2 + [(1, 1, 0)] * 3 + [(1, 2, 0), (7, 2, 3), (7, 1, 0), (3, 1, 0), (5, 1, 2)])
    N = r.choice([1, 1, 2, 3, 4, 8])
    if K == 7 and stride == 1:
        H = W = 7
    else:
        H, W = r.randint(max(3, K), 70), r.randint(max(3, K), 90)
    C1 = r.choice(CH_IN)
    up1 = K == 3 and stride == 1 and pad == 1 and r.random() < 0.25
    C2 = r.choice([0, 0, 32, 64, 128]) if up1 or r.random() < 0.1 else 0
    if C2 and not up1 and not (K == 3 and stride == 1 and pad == 1):
        C2 = 0
    if C2 and C1 % 32:                      # the two-source gather takes multiples of 32 channels per source (a K tile holds one source)
        C1 = (C1 + 31) // 32 * 32
    if up1:
        H, W = max(2, H // 2), max(2, W // 2)
    Cout = r.choice(CH_OUT)
    while N * H * W * (4 if up1 else 1) * max(C1 + C2, Cout) > 40_000_000:
        H, W = max(K, H // 2), max(K, W // 2)
    if r.random() < 0.12:                   # the register-resident 64 -> 64 kernel's domain (conv3x3_c64.hip)
        K, stride, pad, up1, C1, C2, Cout = 3, 1, 1, False, 64, 0, 64
    elif r.random() < 0.06:                 # conv3x3_c32to128.hip
        K, stride, pad, up1, C1, C2, Cout = 3, 1, 1, False, 32, 0, 128
    elif r.random() < 0.06:                 # conv3x3_cat128to32.hip (upsample + concat)
        K, stride, pad, up1, C1, C2, Cout = 3, 1, 1, True, 64, 64, 32
        H, W = max(2, min(H, 40)), max(2, min(W, 60))
    elif r.random() < 0.05:                 # conv7x7s2_stem.hip
        K, stride, pad, up1, C1, C2, Cout = 7, 2, 3, False, 8, 0, 64
    return dict(N=N, H=H, W=W, C1=C1, C2=C2, Cout=Cout, K=K, stride=stride, pad=pad, up1=up1,
                act=r.choice([0, 1, 1, 2]), bias=r.random() < 0.5, res=r.random() < 0.3, stats=r.random() < 0.5)


def run_forward(ops, c, gen):
    N, H, W, C1, C2, Cout, K = c["N"], c["H"], c["W"], c["C1"], c["C2"], c["Cout"], c["K"]
    Hin, Win = (2 * H, 2 * W) if c["up1"] else (H, W)
    x = rnd(gen, N, H, W, C1)
    x2 = rnd(gen, N, Hin, Win, C2) if C2 else None
    Kt = K * K * (C1 + C2)
    w = rnd(gen, Cout, Kt, scale=1.0 / math.sqrt(Kt))
    bias = torch.randn(Cout, generator=gen, device="cuda") if c["bias"] else None
    Ho, Wo = ops.conv_out_size(Hin, K, c["stride"], c["pad"]), ops.conv_out_size(Win, K, c["stride"], c["pad"])
    res = rnd(gen, N, Ho, Wo, Cout) if c["res"] else None
    w_oihw = w.float().view(Cout, K, K, C1 + C2).permute(0, 3, 1, 2).contiguous()
    pre = F.conv2d(gathered_input(x, x2, c["up1"]), w_oihw, bias, stride=c["stride"], padding=c["pad"])
    if res is not None:
        pre = pre + nchw(res)
    want = pre.clamp_min(0) if c["act"] == 1 else torch.sigmoid(pre) if c["act"] == 2 else pre
    out = ops.conv2d(x, w, K, K, x2=x2, bias=bias, res=res, stride=c["stride"], pad=c["pad"], up1=c["up1"], act=c["act"],
                     want_stats=c["stats"])
    got, stats = out if c["stats"] else (out, None)
    torch.cuda.synchronize()
    nbad, emax = err_of(nchw(got), want, 4e-3, 2e-3 + 1e-3 * (1 if c["res"] or c["bias"] else 0))
    msg = "" if nbad == 0 else "forward: %d elements off, max err %.4g" % (nbad, emax)
    if stats is not None and not msg:
        # BatchNorm partial sums are those of the fp16-rounded PRE-activation (bias and residual included)
        s = stats.sum(dim=0)
        g32 = got.float() if c["act"] == 0 else pre.permute(0, 2, 3, 1).half().float()
        w1, w2 = g32.sum(dim=(0, 1, 2)), (g32 * g32).sum(dim=(0, 1, 2))
        npix = N * Ho * Wo
        if not torch.allclose(s[0], w1, rtol=2e-3, atol=2e-3 * npix ** 0.5 + 1e-2):
            msg = "stats: sum off by %.4g" % float((s[0] - w1).abs().max())
        elif not torch.allclose(s[1], w2, rtol=3e-3, atol=1e-2 + 1e-3 * npix ** 0.5):
            msg = "stats: sum of squares off by %.4g" % float((s[1] - w2).abs().max())
    return msg


def run_gemm8(ops, c, gen):
    """The large-tile GEMM path (gemm_w8.hip) forced on every case that is a plain GEMM: both tiles against the implicit-GEMM family
    bit for bit and against ATen."""
    if c["K"] != 1 or c["stride"] != 1 or c["pad"] != 0 or c["up1"] or c["C2"] or c["C1"] % 64 or c["Cout"] % 8 or c["act"] == 2:
        return None
    from hallucidet_amd import _abi
    lib = _abi.load()
    N, H, W, C1, Cout = c["N"], c["H"], c["W"], c["C1"], c["Cout"]
    x = rnd(gen, N, H, W, C1)
    w = rnd(gen, Cout, C1, scale=1.0 / math.sqrt(C1))
    bias = torch.randn(Cout, generator=gen, device="cuda") if c["bias"] else None
    mask = (torch.rand(N, H, W, Cout, generator=gen, device="cuda") > 0.4).half() if c["res"] else None
    kw = dict(bias=bias, mask=mask, act=c["act"])
    try:
        lib.hd_gemm_w8_mode(0)
        ref = ops.conv2d(x, w, 1, 1, **kw)
        outs = []
        for bn in (128, 1128):
            lib.hd_gemm_w8_mode(bn)
            outs.append(ops.conv2d(x, w, 1, 1, **kw))
        torch.cuda.synchronize()
    finally:
        lib.hd_gemm_w8_mode(-1)
    for bn, o in zip((128, 1128), outs):
        if not torch.equal(o, ref):
            return "gemm_w8 tile %d differs from the igemm family (%d elements)" % (bn, int((o != ref).sum()))
    want = x.reshape(-1, C1).float() @ w.float().t()
    if bias is not None:
        want = want + bias
    if mask is not None:
        want = want * (mask.reshape(want.shape).float() > 0)
    if c["act"] == 1:
        want = want.clamp_min(0)
    nbad, emax = err_of(outs[0].reshape(want.shape), want, 4e-3, 3e-3)
    return "" if nbad == 0 else "gemm_w8: %d elements off, max err %.4g" % (nbad, emax)


def run_w8_tiles(ops, c, gen):
    """Every tile of the 8-wave patch-staged 3x3 family (conv3x3_w8.hip), sub-step split (10-13) and step split (15-17), forced on the cases
    the family is eligible for (3x3 / s1 / p1, channel counts of both sources multiples of 64, Cout % 8 == 0): each against ATen."""
    if c["K"] != 3 or c["stride"] != 1 or c["pad"] != 1 or c["C1"] % 64 or c["C2"] % 64 or c["Cout"] % 8 or c["up1"] != bool(c["C2"]):
        return None
    from hallucidet_amd import _abi
    lib = _abi.load()
    N, H, W, C1, C2, Cout = c["N"], c["H"], c["W"], c["C1"], c["C2"], c["Cout"]
    Hin, Win = (2 * H, 2 * W) if c["up1"] else (H, W)
    x = rnd(gen, N, H, W, C1)
    x2 = rnd(gen, N, Hin, Win, C2) if C2 else None
    Kt = 9 * (C1 + C2)
    w = rnd(gen, Cout, Kt, scale=1.0 / math.sqrt(Kt))
    bias = torch.randn(Cout, generator=gen, device="cuda") if c["bias"] else None
    res = rnd(gen, N, Hin, Win, Cout) if c["res"] else None
    w_oihw = w.float().view(Cout, 3, 3, C1 + C2).permute(0, 3, 1, 2).contiguous()
    pre = F.conv2d(gathered_input(x, x2, c["up1"]), w_oihw, bias, padding=1)
    if res is not None:
        pre = pre + nchw(res)
    want = pre.clamp_min(0) if c["act"] == 1 else torch.sigmoid(pre) if c["act"] == 2 else pre
    msgs = []
    try:
        for cfg in (10, 11, 12, 13, 15, 16, 17, 18, 19, 20):      # 18 / 19 / 20: the 160- / 320- / 96-pixel x 64-channel tiles (conv3x3_m160.hip)
            lib.hd_conv_tune_w8(cfg, 1)
            got, stats = ops.conv2d(x, w, 3, 3, x2=x2, bias=bias, res=res, pad=1, up1=c["up1"], act=c["act"], want_stats=True)
            torch.cuda.synchronize()
            nbad, emax = err_of(nchw(got), want, 4e-3, 2e-3 + 1e-3 * (1 if c["res"] or c["bias"] else 0))
            if nbad:
                msgs.append("tile %d: %d elements off, max err %.4g" % (cfg, nbad, emax))
                continue
            s = stats.sum(dim=0)
            g32 = got.float() if c["act"] == 0 else pre.permute(0, 2, 3, 1).half().float()
            w1, w2 = g32.sum(dim=(0, 1, 2)), (g32 * g32).sum(dim=(0, 1, 2))
            npix = N * Hin * Win
            if not torch.allclose(s[0], w1, rtol=2e-3, atol=2e-3 * npix ** 0.5 + 1e-2):
                msgs.append("tile %d: stats sum off by %.4g" % (cfg, float((s[0] - w1).abs().max())))
            elif not torch.allclose(s[1], w2, rtol=3e-3, atol=1e-2 + 1e-3 * npix ** 0.5):
                msgs.append("tile %d: stats sum of squares off by %.4g" % (cfg, float((s[1] - w2).abs().max())))
    finally:
        lib.hd_conv_tune_w8(-1, 1)
    return "; ".join(msgs)


def run_dgrad(ops, c, gen):
    if c["up1"] or c["C2"]:
        return None
    N, H, W, Cin, Cout, K, stride, pad = c["N"], c["H"], c["W"], c["C1"], c["Cout"], c["K"], c["stride"], c["pad"]
    Cin = random.Random(c["N"] * 7 + H).choice([3, Cin, Cin])          # the stem's 3 input channels now and then
    w_oihw = torch.randn(Cout, Cin, K, K, generator=gen, device="cuda") / math.sqrt(K * K * Cin)
    Ho, Wo = ops.conv_out_size(H, K, stride, pad), ops.conv_out_size(W, K, stride, pad)
    cout_p, cin_p = (Cout + 7) // 8 * 8, (Cin + 7) // 8 * 8
    dy = torch.zeros(N, Ho, Wo, cout_p, dtype=torch.float16, device="cuda")
    dy[..., :Cout] = rnd(gen, N, Ho, Wo, Cout)
    _, wd = ops.weight_prep(w_oihw, want_fwd=True, want_dgrad=True)
    want = torch.nn.grad.conv2d_input((N, Cin, H, W), w_oihw.half().float(), nchw(dy[..., :Cout]), stride=stride, padding=pad)
    got = ops.conv2d(dy, wd, K, K, stride=1, pad=K - 1 - pad, in_dil=stride, out_hw=(H, W), cout=cin_p)
    torch.cuda.synchronize()
    nbad, emax = err_of(nchw(got[..., :Cin]), want, 4e-3, 2e-3)
    if nbad:
        return "dgrad (Cin %d): %d elements off, max err %.4g" % (Cin, nbad, emax)
    if cin_p > Cin and bool((got[..., Cin:] != 0).any()):
        return "dgrad: padding channels not zero"
    return ""


def run_wgrad(ops, c, gen, r):
    N, H, W, C1, C2, Cout, K = c["N"], c["H"], c["W"], c["C1"], c["C2"], c["Cout"], c["K"]
    if K == 7 and c["stride"] == 1:
        return None
    Cout = (Cout + 7) // 8 * 8              # hd_wgrad takes dY with its channels padded to 8 (the head's 3 are stored as 8)
    Hin, Win = (2 * H, 2 * W) if c["up1"] else (H, W)
    x = rnd(gen, N, H, W, C1)
    x2 = rnd(gen, N, Hin, Win, C2) if C2 else None
    Ho, Wo = ops.conv_out_size(Hin, K, c["stride"], c["pad"]), ops.conv_out_size(Win, K, c["stride"], c["pad"])
    dy = rnd(gen, N, Ho, Wo, Cout)
    Cin = C1 + C2
    want = torch.nn.grad.conv2d_weight(gathered_input(x, x2, c["up1"]), (Cout, Cin, K, K), nchw(dy), stride=c["stride"], padding=c["pad"])
    nsplit = r.choice([None, None, 1, 2, 3, 7, 40])
    slab = ops.wgrad(x, dy, K, K, x2=x2, stride=c["stride"], pad=c["pad"], up1=c["up1"], nsplit=nsplit)
    dw = torch.empty(Cout, Cin, K, K, device="cuda")
    ops.wgrad_reduce(slab, dw, K, K, Cin, scale=1.0)
    torch.cuda.synchronize()
    npix = N * Ho * Wo
    nbad, emax = err_of(dw, want, 2e-3, 2e-3 * math.sqrt(npix))
    return "" if nbad == 0 else "wgrad (nsplit %s -> %d): %d elements off, max err %.4g" % (nsplit, slab.shape[0], nbad, emax)


def run_fused(ops, c, gen):
    """hd_conv2d_wgrad against its two separate launches: bit for bit."""
    if not (c["K"] == 3 and c["stride"] == 1 and c["pad"] == 1 and not c["up1"] and not c["C2"]):
        return None
    N, H, W, Cin, Cout = c["N"], c["H"], c["W"], c["C1"], (c["Cout"] + 7) // 8 * 8
    x, dy = rnd(gen, N, H, W, Cin), rnd(gen, N, H, W, Cout)
    w_oihw = torch.randn(Cout, Cin, 3, 3, generator=gen, device="cuda") / math.sqrt(9 * Cin)
    _, wd = ops.weight_prep(w_oihw, want_fwd=True, want_dgrad=True)
    kw = dict(stride=1, pad=1, cout=Cin)
    slab0 = ops.wgrad(x, dy, 3, 3, stride=1, pad=1)
    dx0 = ops.conv2d(dy, wd, 3, 3, **kw)
    slab1, dx1 = ops.wgrad_dgrad(x, dy, 3, 3, wd, stride=1, pad=1, dgrad=kw)
    torch.cuda.synchronize()
    if not torch.equal(dx0, dx1):
        return "fused dgrad+wgrad: data gradient differs from the separate launch"
    if slab0.shape != slab1.shape or not torch.equal(slab0, slab1):
        return "fused dgrad+wgrad: slab differs from the separate launch"
    return ""


def run_bstat(ops, r, gen):
    """hd_conv_args.bs_*: the BatchNorm backward sums emitted by a 3x3 data gradient against hd_bn_bwd_reduce on the stored gradient;
    the gradient itself bit-identical to the plain call."""
    from hallucidet_amd import _abi
    Cc = r.choice([64, 64, 128, 256])
    N, H, W = r.choice([1, 2, 4, 8]), r.randint(6, 70), r.randint(6, 90)
    while N * H * W * Cc > 20_000_000:
        H, W = max(6, H // 2), max(6, W // 2)
    use_z, res = r.random() < 0.5, r.random() < 0.5
    dyv, wd = rnd(gen, N, H, W, Cc, scale=0.3), (torch.randn(Cc, 9 * Cc, generator=gen, device="cuda") / math.sqrt(9 * Cc)).half()
    y_u = rnd(gen, N, H, W, Cc)
    z_u = torch.relu(rnd(gen, N, H, W, Cc)) if use_z else None
    rr = rnd(gen, N, H, W, Cc, scale=0.2) if res else None
    mean, invstd = torch.randn(Cc, generator=gen, device="cuda") * 0.1, torch.rand(Cc, generator=gen, device="cuda") + 0.5
    gamma, beta = torch.rand(Cc, generator=gen, device="cuda") + 0.5, torch.randn(Cc, generator=gen, device="cuda") * 0.2
    bs = dict(y=y_u, z=z_u, mean=mean, invstd=invstd, gamma=gamma, beta=beta, relu=r.random() < 0.85)
    dz = ops.conv2d(dyv, wd, 3, 3, pad=1, res=rr, bstat=bs)
    if bs["part"] is None:
        return None                        # this shape is not routed to a kernel with the sums (the caller keeps hd_bn_bwd_reduce)
    if not torch.equal(dz, ops.conv2d(dyv, wd, 3, 3, pad=1, res=rr)):
        return "bstat: the data gradient differs from the plain call"
    part = torch.empty(64, 2 * Cc, device="cuda")
    ops.check(_abi.load().hd_bn_bwd_reduce(ops.ptr(dz), ops.ptr(z_u), ops.ptr(y_u), ops.ptr(mean), ops.ptr(invstd), ops.ptr(gamma), ops.ptr(beta),
                                           ops.ptr(part), 64, N * H * W, Cc, 1 if bs["relu"] else 0, ops._stream()), "hd_bn_bwd_reduce")
    got, want = bs["part"].double().sum(0), part.double().sum(0)
    e = float((got - want).abs().max()) / max(float(want.abs().max()), 1e-6)
    return "" if e <= 5e-5 else "bstat: sums differ from hd_bn_bwd_reduce by %.2e of the largest" % e


def run_stem_subpixel(ops, r, gen):
    """hd_conv7x7s2_dgrad_thin against hd_conv2d's in_dil = 2 route for the same 7x7 / stride-2 data gradient."""
    N, H, W = r.choice([1, 2, 3, 8]), r.randint(8, 320), r.randint(8, 320)
    Hl, Wl = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    w = torch.randn(64, 3, 7, 7, generator=gen, device="cuda") * 0.05
    wf, wd = ops.weight_prep(w, cin_pad=8, cout_pad=64, want_fwd=True, want_dgrad=True)
    dy, z = rnd(gen, N, Hl, Wl, 64, scale=0.5), torch.relu(rnd(gen, N, Hl, Wl, 64))
    got = ops.conv7x7s2_dgrad_thin(dy, ops.stem_dgrad_weights(wf, 8), (H, W), mask_z=z)
    old = ops.conv2d(ops.relu_bwd(dy, z), wd, 7, 7, stride=1, pad=3, in_dil=2, out_hw=(H, W), cout=8)
    e = float((got.float() - old.float()).abs().max())
    return "" if e <= 2e-3 * float(old.float().abs().max()) + 1e-3 else "stem sub-pixel gradient differs by %.3g" % e


def run_consumer_bn(ops, r, gen):
    """in_scale / in_shift (the producer's BatchNorm + ReLU applied while the small-channel kernels stage their operand) against
    hd_bn_apply followed by the plain call: outputs, BatchNorm partial sums and weight-gradient slabs bit for bit."""
    cin, cout, up = r.choice([8, 16, 32]), r.choice([16, 32]), r.random() < 0.3
    N, H, W = r.choice([1, 2, 3]), r.randint(3, 60), r.randint(3, 90)
    y_raw = rnd(gen, N, H, W, cin, scale=2.0)
    scale = torch.rand(cin, generator=gen, device="cuda") + 0.5
    shift = torch.randn(cin, generator=gen, device="cuda") * 0.7 + 0.3
    w = rnd(gen, cout, 9 * cin, scale=0.1)
    relu = r.random() < 0.8
    z = ops.bn_apply(y_raw, scale, shift, relu=relu)
    a, sa = ops.conv2d(z, w, 3, 3, pad=1, up1=up, want_stats=True)
    b, sb = ops.conv2d(y_raw, w, 3, 3, pad=1, up1=up, want_stats=True, in_scale=scale, in_shift=shift, in_relu=relu)
    torch.cuda.synchronize()
    if not (torch.equal(a, b) and torch.equal(sa, sb)):
        return "consumer-side BN: forward differs (%d->%d, up %s, N %d, %dx%d, relu %s)" % (cin, cout, up, N, H, W, relu)
    if cin >= 16:
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        dy = rnd(gen, N, Ho, Wo, cout)
        ns = r.choice([1, 3, 7, 64])
        s0 = ops.wgrad(z, dy, 3, 3, pad=1, up1=up, nsplit=ns)
        s1 = ops.wgrad(y_raw, dy, 3, 3, pad=1, up1=up, nsplit=ns, in_scale=scale, in_shift=shift, in_relu=relu)
        torch.cuda.synchronize()
        if not torch.equal(s0, s1):
            return "consumer-side BN: weight-gradient slab differs (%d->%d, up %s, N %d, %dx%d, nsplit %d)" % (cin, cout, up, N, H, W, ns)
    return ""


def run_multi(ops, r, gen):
    """hd_conv2d_multi on 2-5 'pyramid levels' of one layer type against the separate calls: bit for bit."""
    K, pad = r.choice([(3, 1), (1, 0)])
    Cin, Cout = r.choice([64, 128, 256]), r.choice([12, 36, 64, 256])
    N = r.choice([1, 2, 8])
    H, W = r.randint(20, 64), r.randint(20, 80)
    w = rnd(gen, Cout, K * K * Cin, scale=1.0 / math.sqrt(K * K * Cin))
    bias = torch.randn(Cout, generator=gen, device="cuda")
    calls = []
    for lvl in range(r.randint(2, 5)):
        calls.append((rnd(gen, N, max(1, H >> lvl), max(1, W >> lvl), Cin), w, K, K, dict(bias=bias, pad=pad, act=r.choice([0, 1]))))
    sep = [ops.conv2d(x, w_, K, K, **kw) for x, w_, _, _, kw in calls]
    got = ops.conv2d_multi(calls)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(sep, got)):
        if not torch.equal(a, b):
            return "multi: level %d differs from its own launch (K %d, %d->%d, N %d, %dx%d)" % (i, K, Cin, Cout, N, H, W)
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-seconds", type=float, default=600.0)
    args = ap.parse_args()
    from hallucidet_amd import ops
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    t0, fails, ran = time.time(), [], {"forward": 0, "dgrad": 0, "wgrad": 0, "fused": 0, "multi": 0, "consumer_bn": 0, "bstat": 0, "stem_subpixel": 0, "gemm8": 0, "w8_tiles": 0}
    for i in range(args.cases):
        if time.time() - t0 > args.max_seconds:
            break
        seed = args.seed * 1_000_003 + i
        r = random.Random(seed)
        gen = torch.Generator(device="cuda").manual_seed(seed)
        c = draw_case(r)
        legs = [("forward", lambda: run_forward(ops, c, gen)), ("dgrad", lambda: run_dgrad(ops, c, gen)),
                ("wgrad", lambda: run_wgrad(ops, c, gen, r)), ("fused", lambda: run_fused(ops, c, gen)),
                ("gemm8", lambda: run_gemm8(ops, c, gen)), ("w8_tiles", lambda: run_w8_tiles(ops, c, gen))]
        if i % 4 == 0:
            legs.append(("multi", lambda: run_multi(ops, r, gen)))
        if i % 4 == 1:
            legs.append(("consumer_bn", lambda: run_consumer_bn(ops, r, gen)))
        if i % 4 == 2:
            legs.append(("bstat", lambda: run_bstat(ops, r, gen)))
        if i % 8 == 3:
            legs.append(("stem_subpixel", lambda: run_stem_subpixel(ops, r, gen)))
        for name, leg in legs:
            try:
                msg = leg()
            except Exception as e:            # a refused shape is a finding too
                msg = "%s: %s" % (type(e).__name__, str(e)[:200])
            if msg is None:
                continue
            ran[name] += 1
            if msg:
                fails.append((seed, name, c, msg))
                print("FAIL seed %d %s %s :: %s" % (seed, name, c, msg), flush=True)
    print("fuzz_conv: %s legs run in %.0f s, %d failures" % (ran, time.time() - t0, len(fails)), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
