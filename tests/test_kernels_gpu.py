"""GPU parity of every HIP kernel against the CPU oracle (oracle/kernels.py), through the C ABI.

Tolerances: operands are rounded to fp16 on both sides, accumulation is fp32 on both sides, so the
difference is summation order plus the final fp16 rounding of the output: |err| <= 2^-10*|y| + K-dependent
noise.  We use rtol=4e-3 / atol scaled with sqrt(K)."""
import math

import pytest
import torch

from oracle import kernels as ok

pytestmark = pytest.mark.gpu


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def close(got, want, rtol=4e-3, atol=2e-3):
    got = got.float().cpu()
    want = want.float()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = (err > tol)
    assert not bad.any(), "max err %.5g (tol %.5g) at %d/%d elements" % (err.max(), tol.flatten()[err.argmax()] if err.numel() else 0, int(bad.sum()), err.numel())


CONV_CASES = [
    # N, H, W, C1, C2, Cout, K, stride, pad, up1, act, bias, res
    (2, 16, 20, 64, 0, 64, 3, 1, 1, False, 0, False, False),
    (2, 16, 20, 64, 0, 128, 3, 2, 1, False, 1, True, True),
    (1, 13, 11, 32, 0, 16, 3, 1, 1, False, 0, False, False),     # ragged M tail, Cout < 32
    (2, 8, 10, 64, 64, 32, 3, 1, 1, True, 0, False, False),      # upsample + concat
    (1, 6, 7, 128, 64, 128, 3, 1, 1, True, 1, False, False),      # upsample + concat, K tiles on both sides of the boundary
    (2, 8, 10, 64, 0, 16, 3, 1, 1, True, 1, False, False),       # upsample only
    (1, 32, 32, 8, 0, 64, 7, 2, 3, False, 0, False, False),      # stem: Cin padded to 8, K tile straddles taps
    (2, 15, 15, 64, 0, 256, 1, 1, 0, False, 1, True, False),     # 1x1
    (2, 15, 15, 128, 0, 256, 1, 2, 0, False, 0, True, False),    # 1x1 stride 2 (odd extent)
    (1, 9, 9, 16, 0, 3, 3, 1, 1, False, 2, True, False),         # head: Cout=3 + bias + sigmoid
    (3, 19, 19, 256, 0, 256, 3, 1, 1, False, 1, True, False),    # detector-like
    (1, 7, 7, 256, 0, 1024, 7, 1, 0, False, 1, True, False),     # fc6 as 7x7 conv over RoI features
    # large grids: every (tile, K-depth, stage-count) variant the dispatcher can pick must be exercised
    (6, 150, 150, 8, 0, 64, 7, 2, 3, False, 1, True, False),     # stem at detector size: 128x64 tile, 32-deep, 2 stages, per-lane taps
    (4, 96, 96, 32, 0, 128, 3, 1, 1, False, 1, False, True),     # 128x128 tile, 32-deep, 2 stages
    (4, 96, 96, 256, 0, 128, 1, 1, 0, False, 0, True, False),    # 128x128 tile, 64-deep, 2 stages (1x1: igemm family)
    (4, 96, 96, 256, 0, 64, 1, 1, 0, False, 1, False, False),     # 128x64 tile, 64-deep, 2 stages
    (4, 96, 96, 64, 0, 128, 3, 2, 1, False, 0, True, False),     # 3x3 stride 2 stays on the igemm family
    # 3x3 / s1 / p1 with >= 64 channels
    (4, 96, 96, 64, 0, 128, 3, 1, 1, False, 0, True, False),     # BN=128, one channel chunk, full tiles
    (4, 96, 96, 64, 0, 64, 3, 1, 1, False, 1, False, False),      # BN=64
    (2, 13, 21, 128, 0, 192, 3, 1, 1, False, 1, True, True),      # ragged tiles in both directions, 2 chunks, partial N tile, residual
    (1, 9, 40, 256, 0, 72, 3, 1, 1, False, 0, False, False),      # 4 chunks, Cout not a multiple of 32
    (3, 5, 5, 512, 0, 512, 3, 1, 1, False, 1, True, False),       # image smaller than a tile, 8 chunks, 4 N tiles
    (2, 40, 40, 64, 64, 64, 3, 1, 1, True, 1, False, False),      # dual source, 64-deep, large grid
    # 16/32-channel 3x3 layers: the direct small-channel kernel (conv3x3_small.hip); also the (1, 13, 11, 32 -> 16) case above
    (2, 24, 40, 16, 0, 16, 3, 1, 1, False, 0, False, False),      # K = 144: two taps per K step, zero K tail
    (1, 17, 70, 16, 0, 32, 3, 1, 1, False, 0, False, False),      # ragged tiles both ways, two cout tiles
    (2, 10, 33, 8, 0, 16, 3, 1, 1, False, 0, False, False),       # Cin = 8: four taps per K step (head data gradient)
    (2, 8, 10, 32, 0, 16, 3, 1, 1, True, 0, False, False),        # nearest-2x upsampled source
    (3, 20, 64, 32, 0, 32, 3, 1, 1, False, 0, False, False),      # one tap per K step
    # the detector's small-grid, long-K stages (56-184 tiles x 16-72 K tiles)
    (8, 10, 10, 512, 0, 512, 3, 1, 1, False, 1, True, True),      # layer4 3x3, bias + residual + ReLU
    (8, 10, 10, 2048, 0, 512, 1, 1, 0, False, 0, True, False),    # layer4 1x1
    (2, 19, 19, 1024, 0, 256, 1, 1, 0, False, 1, False, False),   # layer3 1x1
    # wide-N GEMM (the box head's fc6 data gradient: K = 1024 -> N = 12 544; 17 x 98 tiles, ragged last M tile)
    (2100, 1, 1, 1024, 0, 12544, 1, 1, 0, False, 0, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward(dev, case):
    from hallucidet_amd import ops
    N, H, W, C1, C2, Cout, K, stride, pad, up1, act, use_bias, use_res = case
    x = rnd(N, H, W, C1, seed=1)
    Hin, Win = (2 * H, 2 * W) if up1 else (H, W)
    x2 = rnd(N, Hin, Win, C2, seed=2) if C2 else None
    Kt = K * K * (C1 + C2)
    w = rnd(Cout, Kt, scale=1.0 / math.sqrt(Kt), seed=3)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(4)) if use_bias else None
    Ho, Wo = ops.conv_out_size(Hin, K, stride, pad), ops.conv_out_size(Win, K, stride, pad)
    res = rnd(N, Ho, Wo, Cout, seed=5) if use_res else None
    want, wstats = ok.conv2d_nhwc(x, w, K, K, x2=x2, bias=bias, res=res, stride=stride, pad=pad, up1=up1, act=act)
    d = lambda t: None if t is None else t.to(dev)
    got, stats = ops.conv2d(d(x), d(w), K, K, x2=d(x2), bias=d(bias), res=d(res), stride=stride, pad=pad, up1=up1,
                            act=act, want_stats=True)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    close(got, want.half())
    s = stats.sum(dim=0).cpu()
    npix = N * Ho * Wo
    assert torch.allclose(s[0], wstats[0], rtol=2e-3, atol=2e-3 * npix ** 0.5 + 1e-2)
    assert torch.allclose(s[1], wstats[1], rtol=3e-3, atol=1e-2)


W8_CASES = [
    # N, H, W, C1, C2, Cout, K, stride, pad, up1, act, bias, res, mask
    (2, 40, 24, 64, 0, 128, 3, 1, 1, False, 1, True, True, False),     # 3x3: both 8-wave families; ragged 8-wide tiles (W = 24 exact, H = 40)
    (3, 19, 21, 128, 0, 192, 3, 1, 1, False, 0, False, False, True),   # ragged in both directions, 2 channel chunks, partial N tile, ReLU mask
    (1, 33, 9, 256, 0, 72, 3, 1, 1, False, 1, True, False, False),     # 4 chunks, Cout not a multiple of 64
    (2, 12, 20, 64, 64, 64, 3, 1, 1, True, 0, False, False, False),    # decoder concat: upsampled + skip source
    (1, 10, 12, 128, 64, 128, 3, 1, 1, True, 1, False, False, False),  # concat with chunks on both sides of the boundary
    (2, 15, 15, 256, 0, 256, 1, 1, 0, False, 1, True, True, False),    # 1x1 (im2col family only)
    (2, 16, 20, 64, 0, 128, 3, 2, 1, False, 1, True, False, False),    # stride 2 (im2col family only)
    (1, 7, 7, 256, 0, 256, 7, 1, 0, False, 1, True, False, False),     # fc6-like 7x7 valid
    (2, 24, 24, 24, 0, 64, 3, 1, 1, False, 0, False, False, False),    # Cin % 64 != 0: per-lane taps (im2col family only)
    (1, 18, 22, 320, 0, 136, 3, 1, 1, False, 0, False, True, False),   # 5 chunks: two whole periods + the odd-chunk tail of the step-split loop
    (1, 16, 20, 512, 0, 128, 3, 1, 1, False, 1, True, False, False),   # 8 chunks = 72 K steps: the layer the step split was built for
]


@pytest.mark.parametrize("cfg", [10, 11, 12, 13, 15, 16, 17, 18, 19, 20])
def test_conv2d_eight_wave_families(dev, cfg):
    """Every tile of the 8-wave patch-staged 3x3 family (conv3x3_w8.hip: cfg 10-13; 15-17 = the step-split main loop of 11-13, shipped
    for 13; 18 / 19 = the 160- / 320-pixel x 64-channel tiles of conv3x3_m160.hip on v_mfma_f32_16x16x32_f16, round 6), forced through hd_conv_tune_w8 wherever it is eligible (the other cases fall through to the 4-wave family), against the
    oracle: outputs, BN partial sums per tile, bias / residual / ReLU-mask / ReLU."""
    from hallucidet_amd import ops, _abi
    lib = _abi.load()
    try:
        for case in W8_CASES:
            N, H, W, C1, C2, Cout, K, stride, pad, up1, act, use_bias, use_res, use_mask = case
            x = rnd(N, H, W, C1, seed=1)
            Hin, Win = (2 * H, 2 * W) if up1 else (H, W)
            x2 = rnd(N, Hin, Win, C2, seed=2) if C2 else None
            Kt = K * K * (C1 + C2)
            w = rnd(Cout, Kt, scale=1.0 / math.sqrt(Kt), seed=3)
            bias = torch.randn(Cout, generator=torch.Generator().manual_seed(4)) if use_bias else None
            Ho, Wo = ops.conv_out_size(Hin, K, stride, pad), ops.conv_out_size(Win, K, stride, pad)
            res = rnd(N, Ho, Wo, Cout, seed=5) if use_res else None
            mask = (torch.rand(N, Ho, Wo, Cout, generator=torch.Generator().manual_seed(6)) > 0.4).half() if use_mask else None
            want, wstats = ok.conv2d_nhwc(x, w, K, K, x2=x2, bias=bias, res=res, stride=stride, pad=pad, up1=up1, act=0)
            if use_mask:
                want = want * mask.float()
                wstats = (want.half().float().sum(dim=(0, 1, 2)), (want.half().float() ** 2).sum(dim=(0, 1, 2)))
            if act == 1:
                want = want.clamp_min(0)
            d = lambda t: None if t is None else t.to(dev)
            for slices in (1,):
                lib.hd_conv_tune_w8(cfg, slices)
                got, stats = ops.conv2d(d(x), d(w), K, K, x2=d(x2), bias=d(bias), res=d(res), mask=d(mask), stride=stride, pad=pad, up1=up1,
                                        act=act, want_stats=True)
                torch.cuda.synchronize()
                close(got, want.half())
                s_ = stats.sum(dim=0).cpu()
                assert torch.allclose(s_[0], wstats[0], rtol=2e-3, atol=2e-3 * (N * Ho * Wo) ** 0.5 + 1e-2), (cfg, slices, case)
                assert torch.allclose(s_[1], wstats[1], rtol=3e-3, atol=1e-2), (cfg, slices, case)
                again, _ = ops.conv2d(d(x), d(w), K, K, x2=d(x2), bias=d(bias), res=d(res), mask=d(mask), stride=stride, pad=pad, up1=up1,
                                      act=act, want_stats=True)
                assert torch.equal(got, again), "run-to-run identical"
    finally:
        lib.hd_conv_tune_w8(-1, 0)


M160_CASES = [
    # N, H, W, C1, C2, Cout, act, bias, res, mask   (3x3 / s1 / p1 through the DEFAULT dispatcher: these maps are exact covers of 4 x 40 tiles)
    (2, 32, 40, 256, 0, 256, 0, False, False, False),      # ResNet-34 layer3 at 256 x 320 input (the 12-GFLOP shape at batch 8)
    (2, 32, 40, 256, 0, 256, 1, True, True, True),         # the same with every epilogue option
    (1, 64, 80, 128, 0, 128, 0, False, True, False),       # layer2: two channel tiles, 32 pixel tiles per image
    (2, 16, 20, 512, 256, 256, 0, False, False, False),    # decoder block 0's conv1: nearest-2x(512 @16x20) ++ skip(256 @32x40), 12 chunks
    (1, 32, 40, 192, 0, 136, 1, False, False, False),      # 3 chunks (odd: the five-round tail), Cout not a multiple of 64
    (1, 8, 40, 64, 0, 128, 0, True, False, False),         # one chunk only (nine K steps: prologue + tail, no whole period)
    (2, 64, 80, 128, 0, 128, 1, True, True, True),         # layer2 with every epilogue option (the 320-pixel tile: two epilogue halves per block)
    (1, 32, 40, 256, 128, 128, 0, False, False, False),    # decoder block 1's conv1: nearest-2x(256 @32x40) ++ skip(128 @64x80)
]


@pytest.mark.parametrize("cfg", [18, 19])
@pytest.mark.parametrize("case", M160_CASES)
def test_conv2d_160_pixel_tile_on_the_unet_maps(dev, case, cfg):
    """conv3x3_m160.hip (round 6; producer / consumer wave roles) on the U-Net's 32x40 / 64x80 maps, which its 4 x 40- (cfg 18) and
    8 x 40-pixel (cfg 19) x 64-channel tiles cover exactly -- forced through hd_conv_tune_w8 (the shipped rule picks them by a cost model
    that sees the batch: next test): outputs, BatchNorm sums (one row per block) and every epilogue option against the oracle, run-to-run
    identical."""
    from hallucidet_amd import ops, _abi
    lib = _abi.load()
    N, H, W, C1, C2, Cout, act, use_bias, use_res, use_mask = case
    up1 = C2 > 0
    x = rnd(N, H, W, C1, seed=1)
    Hin, Win = (2 * H, 2 * W) if up1 else (H, W)
    x2 = rnd(N, Hin, Win, C2, seed=2) if C2 else None
    Kt = 9 * (C1 + C2)
    w = rnd(Cout, Kt, scale=1.0 / math.sqrt(Kt), seed=3)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(4)) if use_bias else None
    res = rnd(N, Hin, Win, Cout, seed=5) if use_res else None
    mask = (torch.rand(N, Hin, Win, Cout, generator=torch.Generator().manual_seed(6)) > 0.4).half() if use_mask else None
    want, wstats = ok.conv2d_nhwc(x, w, 3, 3, x2=x2, bias=bias, res=res, pad=1, up1=up1, act=0)
    if use_mask:
        want = want * mask.float()
        wstats = (want.half().float().sum(dim=(0, 1, 2)), (want.half().float() ** 2).sum(dim=(0, 1, 2)))
    if act == 1:
        want = want.clamp_min(0)
    d = lambda t: None if t is None else t.to(dev)
    try:
        lib.hd_conv_tune_w8(cfg, 1)
        got, stats = ops.conv2d(d(x), d(w), 3, 3, x2=d(x2), bias=d(bias), res=d(res), mask=d(mask), pad=1, up1=up1, act=act, want_stats=True)
        again, _ = ops.conv2d(d(x), d(w), 3, 3, x2=d(x2), bias=d(bias), res=d(res), mask=d(mask), pad=1, up1=up1, act=act, want_stats=True)
    finally:
        lib.hd_conv_tune_w8(-1, 0)
    torch.cuda.synchronize()
    th = 4 if cfg == 18 else 8
    assert stats.shape[0] == N * ((Hin + th - 1) // th) * (Win // 40), "not the %d x 40-pixel tile (%d rows)" % (th, stats.shape[0])
    close(got, want.half())
    s_ = stats.sum(dim=0).cpu()
    assert torch.allclose(s_[0], wstats[0], rtol=2e-3, atol=2e-3 * (N * Hin * Win) ** 0.5 + 1e-2)
    assert torch.allclose(s_[1], wstats[1], rtol=3e-3, atol=1e-2)
    assert torch.equal(got, again), "run-to-run identical"


M96_CASES = [
    # N, H, W, Cin, Cout, act, bias, res, mask   (3x3 / s1 / p1; the 4 x 24-pixel x 64-channel tile, cfg 20)
    (3, 19, 19, 256, 256, 1, True, False, False),          # the detector's layer3 bottleneck conv2: ragged in both directions (19 = 4 * 4 + 3, 24 - 5)
    (2, 19, 19, 256, 256, 0, False, True, True),           # its data gradient form: residual + ReLU mask
    (2, 10, 10, 512, 512, 1, True, False, False),          # layer4: 8 chunks, 10 of 24 columns live
    (1, 38, 38, 128, 136, 0, False, False, False),         # two column tiles (24 + 14), Cout not a multiple of 64
    (1, 8, 48, 64, 128, 1, True, True, False),             # exact cover, one chunk only (prologue + tail)
    (2, 5, 5, 192, 128, 0, False, False, False),           # the P6-sized map, 3 chunks (odd tail)
]


@pytest.mark.parametrize("case", M96_CASES)
def test_conv2d_96_pixel_tile_on_the_detector_maps(dev, case):
    """conv3x3_m160.hip's 4 x 24-pixel x 64-channel instance (cfg 20; three 8-column blocks per row pair, same producer / consumer roles and
    swizzle) forced through hd_conv_tune_w8 on the detector's small maps: outputs, one BatchNorm row per block, every epilogue option against
    the oracle, run-to-run identical."""
    from hallucidet_amd import ops, _abi
    lib = _abi.load()
    N, H, W, Cin, Cout, act, use_bias, use_res, use_mask = case
    x = rnd(N, H, W, Cin, seed=1)
    w = rnd(Cout, 9 * Cin, scale=1.0 / math.sqrt(9 * Cin), seed=3)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(4)) if use_bias else None
    res = rnd(N, H, W, Cout, seed=5) if use_res else None
    mask = (torch.rand(N, H, W, Cout, generator=torch.Generator().manual_seed(6)) > 0.4).half() if use_mask else None
    want, wstats = ok.conv2d_nhwc(x, w, 3, 3, bias=bias, res=res, pad=1, act=0)
    if use_mask:
        want = want * mask.float()
        wstats = (want.half().float().sum(dim=(0, 1, 2)), (want.half().float() ** 2).sum(dim=(0, 1, 2)))
    if act == 1:
        want = want.clamp_min(0)
    d = lambda t: None if t is None else t.to(dev)
    try:
        lib.hd_conv_tune_w8(20, 1)
        got, stats = ops.conv2d(d(x), d(w), 3, 3, bias=d(bias), res=d(res), mask=d(mask), pad=1, act=act, want_stats=True)
        again, _ = ops.conv2d(d(x), d(w), 3, 3, bias=d(bias), res=d(res), mask=d(mask), pad=1, act=act, want_stats=True)
    finally:
        lib.hd_conv_tune_w8(-1, 0)
    torch.cuda.synchronize()
    assert stats.shape[0] == N * ((H + 3) // 4) * ((W + 23) // 24), "not the 4 x 24-pixel tile (%d rows)" % stats.shape[0]
    close(got, want.half())
    s_ = stats.sum(dim=0).cpu()
    assert torch.allclose(s_[0], wstats[0], rtol=2e-3, atol=2e-3 * (N * H * W) ** 0.5 + 1e-2)
    assert torch.allclose(s_[1], wstats[1], rtol=3e-3, atol=1e-2)
    assert torch.equal(got, again), "run-to-run identical"


def test_conv_dispatcher_routes_the_unet_layers_to_the_160_and_320_pixel_tiles(dev):
    """The shipped rule at the training batch (hd_conv2d_stats_rows answers without launching: one BatchNorm row per block): ResNet-34
    layer3 (8 x 32 x 40 x 256) on 4 x 40-pixel tiles -- 64 x 4 = 256 blocks --, layer2 (8 x 64 x 80 x 128) on 8 x 40-pixel tiles -- 128 x 2 =
    256 blocks --, decoder block 0's conv1 likewise; a single image goes to fewer, cheaper blocks; the detector's 19 x 19 maps to 4 x 24-pixel tiles."""
    import ctypes as C
    from hallucidet_amd import _abi
    lib = _abi.load()

    def rows(N, H, W, Cin, Cout):
        a = _abi.ConvArgs(x=8, w=8, y=8, N=N, Hsrc=H, Wsrc=W, Hin=H, Win=W, C1=Cin, C2=0, Ho=H, Wo=W, Cout=Cout, KH=3, KW=3, stride=1, pad=1,
                          up1=0, in_dil=1, act=0, out_mode=0)
        return lib.hd_conv2d_stats_rows(C.byref(a))
    assert rows(8, 32, 40, 256, 256) == 8 * 8 * 1
    assert rows(8, 64, 80, 128, 128) == 8 * 8 * 2
    assert rows(1, 32, 40, 256, 256) in (1 * 2 * 5, 1 * 1 * 5, 1 * 8 * 2)   # 16 x 8- or 32 x 8-pixel tiles, or the 4 x 24-pixel tile (64 blocks, two per CU)
    assert rows(24, 19, 19, 256, 256) == 24 * 5 * 1                     # the detector's layer3 at batch 24: 4 x 24-pixel tiles, 480 blocks in co-resident pairs
    assert rows(8, 19, 19, 256, 256) == 8 * 5 * 1


@pytest.mark.parametrize("cfg", [18, 19, 20])
@pytest.mark.parametrize("case", [(1, 8, 40, 128, 128, 64), (2, 32, 40, 256, 256, 128), (1, 12, 80, 128, 64, 0), (1, 10, 44, 64, 128, 64)])
def test_conv2d_160_pixel_tile_pooled_half_and_skip_half(dev, case, cfg):
    """out_pool2 on the 160- / 320-pixel tiles (forced through hd_conv_tune_w8(18 / 19): the shared epilogue's 2 x 2 sum with a 40-pixel
    tile pitch): pooled half against the plain call + hd_concat_up_bwd and the oracle, skip half bit for bit; ragged maps."""
    from hallucidet_amd import ops, _abi
    lib = _abi.load()
    N, H, W, Cin, c_up, c_skip = case
    Cout = c_up + c_skip
    x = rnd(N, H, W, Cin, seed=1).to(dev)
    w = rnd(Cout, 9 * Cin, scale=1.0 / math.sqrt(9 * Cin), seed=3).to(dev)
    try:
        lib.hd_conv_tune_w8(cfg, 1)
        info = dict(c_up=c_up)
        got = ops.conv2d(x, w, 3, 3, pad=1, pool2=info)
        assert info["done"] and got.shape == (N, H // 2, W // 2, c_up)
        plain = ops.conv2d(x, w, 3, 3, pad=1)
    finally:
        lib.hd_conv_tune_w8(-1, 0)
    da, ds = ops.concat_up_bwd(plain, c_up)
    torch.cuda.synchronize()
    if c_skip:
        assert torch.equal(info["skip"], ds)
    else:
        assert info["skip"] is None and ds is None
    assert float((got.float() - da.float()).abs().max()) <= 4e-3 * max(1.0, float(da.float().abs().max()))
    want, _ = ok.conv2d_nhwc(x.cpu(), w.cpu(), 3, 3, pad=1)
    close(got, want[..., :c_up].reshape(N, H // 2, 2, W // 2, 2, c_up).sum(dim=(2, 4)).half())
    close(plain, want.half())


GEMM8_CASES = [
    # N, H, W, Cin, Cout, K (window = map), bias + ReLU, mask
    (700, 1, 1, 1024, 1000, 1, True, False),        # FC: ragged last row tile (700 = 2 x 256 + 188), ragged channel tile (1000)
    (300, 7, 7, 256, 1024, 7, True, False),         # fc6: the 7 x 7 window over a 7 x 7 map is the stored row (K = 12 544: 196 K steps)
    (513, 1, 1, 1024, 12544, 1, False, False),      # fc6's data gradient: 49 / 98 channel tiles, one row past two tiles
    (2, 19, 21, 512, 256, 1, False, True),          # 1x1 / stride-1 convolution with the ReLU-backward mask
    (3, 10, 10, 2048, 520, 1, True, True),          # Cout % 8 == 0 only: a channel tile with 8 live channels
    (1, 1, 1, 64, 8, 1, False, False),              # one row, one 8-channel group, one K step
]


@pytest.mark.parametrize("bn", [128, 1128])
def test_gemm_w8_matches_the_igemm_family_bit_for_bit(dev, bn):
    """gemm_w8.hip (256 x 128 tiles on 8 waves / 128 x 128 on 4, register-only epilogue) forced through hd_gemm_w8_mode wherever hd_conv2d's problem is a
    plain GEMM over stored tensors, against (a) the oracle and (b) the 4-wave implicit-GEMM family on the same inputs BIT FOR BIT: the
    products and the fp32 summation order over K are the same, so which path a problem takes may depend on the batch size without
    breaking batch invariance (image n of a batch == the image alone: asserted here across the two paths)."""
    from hallucidet_amd import ops, _abi
    lib = _abi.load()
    try:
        for N, H, W, Cin, Cout, K, bias_relu, use_mask in GEMM8_CASES:
            x = rnd(N, H, W, Cin, seed=1).to(dev)
            Kt = K * K * Cin
            w = rnd(Cout, Kt, scale=1.0 / math.sqrt(Kt), seed=3).to(dev)
            Ho, Wo = (1, 1) if K > 1 else (H, W)
            bias = torch.randn(Cout, generator=torch.Generator().manual_seed(4)).to(dev) if bias_relu else None
            mask = (torch.rand(N, Ho, Wo, Cout, generator=torch.Generator().manual_seed(6)) > 0.4).half().to(dev) if use_mask else None
            kw = dict(bias=bias, mask=mask, act=1 if bias_relu else 0)
            lib.hd_gemm_w8_mode(0)
            ref = ops.conv2d(x, w, K, K, **kw)
            lib.hd_gemm_w8_mode(bn)
            got = ops.conv2d(x, w, K, K, **kw)
            again = ops.conv2d(x, w, K, K, **kw)
            one = ops.conv2d(x[N // 2:N // 2 + 1].contiguous(), w, K, K, bias=bias, act=kw["act"],
                             mask=None if mask is None else mask[N // 2:N // 2 + 1].contiguous())
            torch.cuda.synchronize()
            assert torch.equal(got, ref), (bn, N, H, W, Cin, Cout)
            assert torch.equal(got, again) and torch.equal(one, ref[N // 2:N // 2 + 1])
            want, _ = ok.conv2d_nhwc(x.cpu(), w.cpu(), K, K, bias=None if bias is None else bias.cpu(), act=0)
            if use_mask:
                want = want * mask.cpu().float()
            if bias_relu:
                want = want.clamp_min(0)
            close(got, want.half())
        # a residual joins in fp32 before the bias, as in the 4-wave family's epilogue (bit-identical); what the path does not
        # implement stays where it was (BatchNorm sums, strides, fp32 outputs) -- same call, same answer as with the path switched off
        x = rnd(600, 1, 1, 512, seed=1).to(dev)
        w = rnd(520, 512, scale=1.0 / math.sqrt(512), seed=3).to(dev)
        res = rnd(600, 1, 1, 520, seed=5).to(dev)
        bias = torch.randn(520, generator=torch.Generator().manual_seed(9)).to(dev)
        lib.hd_gemm_w8_mode(bn)
        r1 = ops.conv2d(x, w, 1, 1, res=res, bias=bias, act=1)
        a, sa = ops.conv2d(x, w, 1, 1, res=res, want_stats=True)
        lib.hd_gemm_w8_mode(0)
        r0 = ops.conv2d(x, w, 1, 1, res=res, bias=bias, act=1)
        b, sb = ops.conv2d(x, w, 1, 1, res=res, want_stats=True)
        assert torch.equal(r1, r0) and torch.equal(a, b) and torch.equal(sa, sb)
    finally:
        lib.hd_gemm_w8_mode(-1)


def test_pad_cast_many_equals_the_single_launches(dev):
    """hd_pad_cast_f32_f16_multi (the ten head gradients of an RPN backward pass in one launch) against hd_pad_cast_f32_f16 per tensor
    and the definition: dense sources, slices of a larger buffer (image stride > H * W * C), an empty tensor, 17 tensors (two launches)."""
    from hallucidet_amd import ops
    g = torch.Generator().manual_seed(5)
    xs, cps = [], []
    for i, (n, H, W, C_, cp) in enumerate([(8, 75, 75, 3, 8), (8, 38, 38, 12, 16), (8, 19, 19, 3, 8), (8, 10, 10, 12, 16), (8, 5, 5, 3, 8), (0, 5, 5, 3, 8), (1, 1, 1, 1, 8)] +
                                          [(2, 3 + k, 4, 5, 8) for k in range(10)]):
        if i % 3 == 1 and n:                      # a slice [:n] of a buffer holding more rows per image
            big = torch.randn(n, (H * W + 7) * C_, generator=g).to(dev)
            x = big[:, :H * W * C_].view(n, H, W, C_)
        else:
            x = torch.randn(n, H, W, C_, generator=g).to(dev)
        xs.append(x)
        cps.append(cp)
    many = ops.pad_cast_f32_f16_many(xs, cps)
    torch.cuda.synchronize()
    assert len(many) == 17
    for x, cp, y in zip(xs, cps, many):
        one = ops.pad_cast_f32_f16(x, cp) if x.numel() else torch.empty((0,) + tuple(x.shape[1:3]) + (cp,), dtype=torch.float16, device=dev)
        assert y.shape == x.shape[:3] + (cp,) and torch.equal(y, one)
        assert torch.equal(y[..., :x.shape[3]], x.half()) and not bool(y[..., x.shape[3]:].any())


@pytest.mark.parametrize("case", [(2, 16, 64, 16, 32), (1, 24, 40, 32, 32), (3, 8, 32, 8, 16), (2, 10, 36, 16, 16)])
def test_conv2d_pooled_output_is_the_2x2_sum_of_the_plain_output(dev, case):
    """hd_conv_args.out_pool2 (the small-channel 3x3 kernel's epilogue forms the 2 x 2 sum of its tile in fp32: the data gradient of a
    decoder block without a skip, decoders/unet/decoder.py:38-41) against the plain call followed by hd_concat_up_bwd -- equal up to the
    one fp16 rounding the fused form does not make (the plain path rounds the four addends first) -- and against the oracle's fp32
    convolution pooled on the host; ragged tiles, every Cin / Cout the kernel has."""
    from hallucidet_amd import ops
    N, H, W, Cin, Cout = case
    x = rnd(N, H, W, Cin, seed=1).to(dev)
    w = rnd(Cout, 9 * Cin, scale=1.0 / math.sqrt(9 * Cin), seed=3).to(dev)
    info = {}
    got = ops.conv2d(x, w, 3, 3, pad=1, pool2=info)
    assert info["done"] and got.shape == (N, H // 2, W // 2, Cout)
    plain = ops.conv2d(x, w, 3, 3, pad=1)
    two, none = ops.concat_up_bwd(plain, Cout)
    torch.cuda.synchronize()
    assert none is None
    want, _ = ok.conv2d_nhwc(x.cpu(), w.cpu(), 3, 3, pad=1)
    want = want.reshape(N, H // 2, 2, W // 2, 2, Cout).sum(dim=(2, 4))
    close(got, want.half())
    assert float((got.float() - two.float()).abs().max()) <= 4e-3 * max(1.0, float(two.float().abs().max()))
    again = ops.conv2d(x, w, 3, 3, pad=1, pool2={})
    assert torch.equal(got, again)
    # not implemented elsewhere: the request is declined (the caller pools), never silently ignored by the kernel
    big = {}
    y = ops.conv2d(rnd(1, 8, 8, 64, seed=2).to(dev), rnd(64, 9 * 64, scale=0.05, seed=4).to(dev), 3, 3, pad=1, pool2=big)
    assert big["done"] is False and y.shape == (1, 8, 8, 64)


@pytest.mark.parametrize("case", [(2, 16, 16, 128, 256, 128), (1, 32, 24, 256, 512, 256), (2, 16, 24, 64, 128, 64), (1, 16, 8, 128, 256, 0)])
def test_conv2d_eight_wave_pooled_half_and_skip_half(dev, case):
    """out_pool2 in the 8-wave 3x3 family (decoder blocks 0-2: the data gradient of conv1 over cat([nearest_2x(a), skip]) with 512 + 256,
    256 + 128, 128 + 64 channels): the upsampled half 2 x 2 sum-pooled, the skip half unpooled in y2 -- against the plain call +
    hd_concat_up_bwd and the oracle; alone and inside the fused data + weight gradient grid (ops.wgrad_dgrad); ragged tiles; a case with
    no skip half."""
    from hallucidet_amd import ops
    N, H, W, Cin, c_up, c_skip = case
    Cout = c_up + c_skip
    x = rnd(N, H, W, Cin, seed=1).to(dev)
    w = rnd(Cout, 9 * Cin, scale=1.0 / math.sqrt(9 * Cin), seed=3).to(dev)
    info = dict(c_up=c_up)
    got = ops.conv2d(x, w, 3, 3, pad=1, pool2=info)
    assert info["done"] and got.shape == (N, H // 2, W // 2, c_up)
    declined = {}                                 # a map the 8-wide tiles cover badly stays on the 4-wave family: the request is declined
    y = ops.conv2d(rnd(1, 12, 20, Cin, seed=5).to(dev), w, 3, 3, pad=1, pool2=dict(declined, c_up=c_up))
    assert y.shape[-1] in (Cout, c_up)
    plain = ops.conv2d(x, w, 3, 3, pad=1)
    da, ds = ops.concat_up_bwd(plain, c_up)
    torch.cuda.synchronize()
    if c_skip:
        assert torch.equal(info["skip"], ds)
    else:
        assert info["skip"] is None and ds is None
    assert float((got.float() - da.float()).abs().max()) <= 4e-3 * max(1.0, float(da.float().abs().max()))
    want, _ = ok.conv2d_nhwc(x.cpu(), w.cpu(), 3, 3, pad=1)
    close(got, want[..., :c_up].reshape(N, H // 2, 2, W // 2, 2, c_up).sum(dim=(2, 4)).half())
    # the same request through the one-grid data + weight gradient call (the convolution's x is the layer's dY there)
    xin = rnd(N, H // 2, W // 2, c_up, seed=7).to(dev)
    skip = rnd(N, H, W, c_skip, seed=8).to(dev) if c_skip else None
    wd = rnd(Cout, 9 * Cin, scale=1.0 / math.sqrt(9 * Cin), seed=9).to(dev)
    info2 = dict(c_up=c_up)
    slab, dx = ops.wgrad_dgrad(xin, x, 3, 3, wd, x2=skip, pad=1, up1=True, dgrad=dict(pad=1, cout=Cout, pool2=info2))
    ref_slab = ops.wgrad(xin, x, 3, 3, x2=skip, pad=1, up1=True)
    ref_dx = ops.conv2d(x, wd, 3, 3, pad=1, cout=Cout, pool2=dict(c_up=c_up))
    torch.cuda.synchronize()
    assert info2["done"] and torch.equal(dx, ref_dx) and torch.equal(slab, ref_slab)


@pytest.mark.parametrize("case", [(2, 16, 32), (1, 24, 48), (3, 10, 20)])
def test_conv2d_c32to128_pooled_half_and_skip_half(dev, case):
    """Decoder block 3's data gradient (32 -> 128 channels, conv3x3_c32to128.hip) with out_pool2 = 64: channels 0..63 leave 2 x 2
    sum-pooled (the gradient of the upsampled source), channels 64..127 unpooled in a second tensor (the skip's gradient) -- against the
    plain call + hd_concat_up_bwd (the skip half bit for bit, the pooled half up to the one rounding the fused form saves) and the
    oracle; ragged 8 x 16 tiles."""
    from hallucidet_amd import ops
    N, H, W = case
    x = rnd(N, H, W, 32, seed=1).to(dev)
    w = rnd(128, 9 * 32, scale=1.0 / math.sqrt(9 * 32), seed=3).to(dev)
    info = dict(c_up=64)
    got = ops.conv2d(x, w, 3, 3, pad=1, pool2=info)
    assert info["done"] and got.shape == (N, H // 2, W // 2, 64) and info["skip"].shape == (N, H, W, 64)
    plain = ops.conv2d(x, w, 3, 3, pad=1)
    da, ds = ops.concat_up_bwd(plain, 64)
    torch.cuda.synchronize()
    assert torch.equal(info["skip"], ds)
    assert float((got.float() - da.float()).abs().max()) <= 4e-3 * max(1.0, float(da.float().abs().max()))
    want, _ = ok.conv2d_nhwc(x.cpu(), w.cpu(), 3, 3, pad=1)
    close(got, want[..., :64].reshape(N, H // 2, 2, W // 2, 2, 64).sum(dim=(2, 4)).half())
    close(info["skip"], want[..., 64:].half())


def test_conv2d_nchw_f32_output(dev):
    from hallucidet_amd import ops
    x = rnd(2, 12, 16, 16, seed=1)
    w = rnd(3, 9 * 16, scale=0.1, seed=2)
    bias = torch.tensor([0.1, -0.2, 0.3])
    want, _ = ok.conv2d_nhwc(x, w, 3, 3, bias=bias, pad=1, act=2)
    got = ops.conv2d(x.to(dev), w.to(dev), 3, 3, bias=bias.to(dev), pad=1, act=2, out_nchw_f32=True)
    assert got.dtype == torch.float32 and got.shape == (2, 3, 12, 16)
    close(got, want.permute(0, 3, 1, 2), rtol=1e-4, atol=1e-5)


DGRAD_CASES = [
    # N, H, W, Cin, Cout, K, stride, pad
    (2, 16, 20, 64, 64, 3, 1, 1),
    (2, 16, 20, 32, 64, 3, 2, 1),
    (2, 15, 15, 64, 128, 3, 2, 1),   # odd extent (detector 75 -> 38)
    (2, 15, 15, 64, 128, 1, 2, 0),
    (1, 30, 30, 3, 64, 7, 2, 3),     # stem, Cin=3 padded to 8
    (2, 9, 9, 16, 3, 3, 1, 1),       # head: Cout=3 padded to 8
    (2, 21, 19, 128, 256, 3, 1, 1),  # input-patch kernel as data gradient: 4 chunks of dy, ragged tiles
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv2d_dgrad(dev, case):
    """Data gradient = the same implicit-GEMM kernel on flipped/transposed weights (+ zero-dilated dY for stride>1)."""
    from hallucidet_amd import ops
    N, H, W, Cin, Cout, K, stride, pad = case
    g = torch.Generator().manual_seed(7)
    w_oihw = torch.randn(Cout, Cin, K, K, generator=g) / math.sqrt(K * K * Cin)
    Ho, Wo = ops.conv_out_size(H, K, stride, pad), ops.conv_out_size(W, K, stride, pad)
    cout_p, cin_p = (Cout + 7) // 8 * 8, (Cin + 7) // 8 * 8
    dy = torch.zeros(N, Ho, Wo, cout_p, dtype=torch.float16)
    dy[..., :Cout] = rnd(N, Ho, Wo, Cout, seed=8)
    wf, wd = ops.weight_prep(w_oihw.to(dev), want_fwd=True, want_dgrad=True)
    torch.cuda.synchronize()
    # oracle uses the fp16-rounded forward weights
    w_flat = w_oihw.half().float().permute(0, 2, 3, 1).reshape(Cout, K * K * Cin)
    want = ok.conv2d_dgrad_nhwc(dy[..., :Cout], w_flat, K, K, Cin, stride=stride, pad=pad, in_hw=(H, W))
    got = ops.conv2d(dy.to(dev), wd, K, K, stride=1, pad=K - 1 - pad, in_dil=stride, out_hw=(H, W), cout=cin_p)
    torch.cuda.synchronize()
    assert got.shape == (N, H, W, cin_p)
    close(got[..., :Cin], want.half())
    if cin_p > Cin:
        assert (got[..., Cin:] == 0).all()
    # forward layout check: wf equals the (kh,kw,ci) flattening with zero channel padding
    wf_want = torch.zeros(Cout, K * K, cin_p)
    wf_want[:, :, :Cin] = w_oihw.permute(0, 2, 3, 1).reshape(Cout, K * K, Cin)
    assert torch.equal(wf.cpu().float(), wf_want.half().float().reshape(Cout, -1))


WGRAD_CASES = [
    # N, H, W, C1, C2, Cout, K, stride, pad, up1, nsplit
    (2, 16, 20, 64, 0, 64, 3, 1, 1, False, 3),
    (2, 16, 20, 32, 0, 128, 3, 2, 1, False, 2),
    (2, 8, 10, 32, 32, 32, 3, 1, 1, True, 4),
    (1, 13, 11, 16, 0, 16, 3, 1, 1, False, 1),
    (1, 32, 32, 8, 0, 64, 7, 2, 3, False, 5),
    (2, 15, 15, 64, 0, 128, 1, 2, 0, False, 2),
    (2, 9, 9, 16, 0, 8, 3, 1, 1, False, 2),
    (2, 8, 8, 512, 0, 512, 3, 1, 1, False, 2),
    # small-channel 3x3 layers: wgrad3x3_small.hip (also the (16 -> 16) and (16 -> 8) cases above)
    (2, 24, 70, 16, 0, 16, 3, 1, 1, False, 7),      # more tiles than blocks, ragged tiles
    (1, 17, 40, 32, 0, 32, 3, 1, 1, False, 3),      # two cout tiles x two cin tiles
    (2, 8, 10, 32, 0, 16, 3, 1, 1, True, 2),        # nearest-2x upsampled source
    (1, 9, 33, 16, 0, 32, 3, 1, 1, False, 50),      # more blocks than tiles (empty partials)
    (3, 16, 64, 32, 0, 24, 3, 1, 1, False, 4),      # Cout not a multiple of 16
    (2, 9, 21, 64, 64, 32, 3, 1, 1, True, 5),       # thin output on the decoder concat (64 upsampled + 64 skip channels -> 32), ragged tiles
    (1, 8, 16, 64, 64, 16, 3, 1, 1, True, 300),     # the same with one cout tile and more blocks than tiles
    # 3x3 / s1 layers with >= 64 channels: the 8-wave patch-staged kernel (wgrad3x3_w8.hip; also the first and the 512-channel cases above)
    (2, 40, 24, 64, 0, 128, 3, 1, 1, False, 5),     # two cout chunks; 16x8 tiles exact in W, ragged in H
    (3, 19, 21, 128, 0, 64, 3, 1, 1, False, 4),     # ragged both ways, two ci chunks
    (1, 33, 9, 192, 0, 64, 3, 1, 1, False, 40),     # more slices than tiles (empty partials must be zero)
    (2, 12, 20, 64, 64, 64, 3, 1, 1, True, 3),      # decoder concat: upsampled source + skip source
    (1, 10, 12, 128, 64, 128, 3, 1, 1, True, 2),    # concat with two chunks in the first source
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_wgrad(dev, case):
    """Transposing-LDS-read (ds_read_b64_tr_b16) weight-gradient GEMM vs torch.nn.grad.conv2d_weight."""
    from hallucidet_amd import ops
    N, H, W, C1, C2, Cout, K, stride, pad, up1, nsplit = case
    x = rnd(N, H, W, C1, seed=11)
    Hin, Win = (2 * H, 2 * W) if up1 else (H, W)
    x2 = rnd(N, Hin, Win, C2, seed=12) if C2 else None
    Ho, Wo = ops.conv_out_size(Hin, K, stride, pad), ops.conv_out_size(Win, K, stride, pad)
    dy = rnd(N, Ho, Wo, Cout, seed=13)
    want = ok.conv2d_wgrad_nhwc(x, dy, K, K, x2=x2, stride=stride, pad=pad, up1=up1)
    d = lambda t: None if t is None else t.to(dev)
    slab = ops.wgrad(d(x), d(dy), K, K, x2=d(x2), stride=stride, pad=pad, up1=up1, nsplit=nsplit)
    torch.cuda.synchronize()
    got = slab.sum(dim=0).cpu()
    npix = N * Ho * Wo
    close(got, want, rtol=2e-3, atol=2e-3 * math.sqrt(npix))
    # reduce kernel -> OIHW
    Cin = C1 + C2
    dw = torch.empty(Cout, Cin, K, K, device=dev)
    ops.wgrad_reduce(slab, dw, K, K, Cin, scale=0.5)
    want_oihw = 0.5 * want.view(Cout, K, K, Cin).permute(0, 3, 1, 2)
    close(dw, want_oihw, rtol=2e-3, atol=2e-3 * math.sqrt(npix))


def test_bn_train_forward_backward(dev):
    from hallucidet_amd import ops
    N, H, W, C = 2, 12, 10, 64
    y = rnd(N, H, W, C, scale=2.0, seed=21) + 0.5
    y = y.half()
    res = rnd(N, H, W, C, seed=22)
    gamma = torch.rand(C) + 0.5
    beta = torch.randn(C) * 0.1
    eps = 1e-5
    # forward via conv-stats path: emulate stats slab from the tensor itself
    yf = y.float()
    sums = torch.stack([yf.sum(dim=(0, 1, 2)), (yf * yf).sum(dim=(0, 1, 2))]).reshape(1, -1)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd, scale, shift = ops.bn_finalize(sums.to(dev).view(1, 2, C), N * H * W, gamma.to(dev), beta.to(dev), rm, rv, 0.1, eps)
    z = ops.bn_apply(y.to(dev), scale, shift, res=res.to(dev), relu=True)
    zw, mw, iw = ok.bn_train_nhwc(y, gamma, beta, eps, res=res, relu=True)
    close(z, zw.half())
    assert torch.allclose(mean.cpu(), mw, atol=1e-4) and torch.allclose(invstd.cpu(), iw, rtol=1e-4)
    # running stats (momentum 0.1, unbiased var)
    n = N * H * W
    assert torch.allclose(rm.cpu(), 0.1 * mw, atol=1e-5)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * yf.var(dim=(0, 1, 2), unbiased=True), rtol=1e-4)
    # backward vs autograd
    yv = y.float().clone().requires_grad_(True)
    rv_ = res.float().clone().requires_grad_(True)
    gv = gamma.clone().requires_grad_(True)
    bv = beta.clone().requires_grad_(True)
    zz, _, _ = ok.bn_train_nhwc(yv, gv, bv, eps, res=rv_, relu=False)
    dz = rnd(N, H, W, C, seed=23)
    # use the HIP z for the mask on both sides to avoid ulp-level mask flips
    mask = (z.cpu().float() > 0).float()
    (zz * mask).backward(dz.float())
    dy, dres, dgamma, dbeta = ops.bn_backward(dz.to(dev), z, y.to(dev), mean, invstd, gamma.to(dev), beta.to(dev), relu=True, want_dres=True, gscale=1.0)
    close(dy, yv.grad.half(), rtol=1e-2, atol=3e-3)
    close(dres, rv_.grad.half())
    assert torch.allclose(dgamma.cpu(), gv.grad, rtol=1e-3, atol=1e-2)
    assert torch.allclose(dbeta.cpu(), bv.grad, rtol=1e-3, atol=1e-2)


def test_bn_eval_scale_shift(dev):
    from hallucidet_amd import ops
    C = 32
    g, b, rm, rv = torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.1
    s, t = ops.bn_eval_scale_shift(g.to(dev), b.to(dev), rm.to(dev), rv.to(dev), 1e-5)
    ws = g / torch.sqrt(rv + 1e-5)
    assert torch.allclose(s.cpu(), ws, rtol=1e-5) and torch.allclose(t.cpu(), b - rm * ws, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("hw", [(16, 20), (15, 15), (7, 9)])
def test_maxpool_fwd_bwd(dev, hw):
    from hallucidet_amd import ops
    H, W = hw
    x = torch.relu(rnd(2, H, W, 16, seed=31)).half()  # post-ReLU: many exact ties at 0
    y = ops.maxpool3x3s2(x.to(dev))
    want = ok.maxpool3x3s2_nhwc(x)
    assert torch.equal(y.cpu().float(), want)
    xv = ok.nhwc_to_nchw(x.float()).requires_grad_(True)
    out = torch.nn.functional.max_pool2d(xv, 3, 2, 1)
    dy = rnd(*y.shape, seed=32)
    out.backward(ok.nhwc_to_nchw(dy.float()))
    dx = ops.maxpool3x3s2_bwd(x.to(dev), dy.to(dev))
    close(dx, ok.nchw_to_nhwc(xv.grad).half(), rtol=2e-3, atol=1e-3)
    # index-recording form: identical output, identical routing (ties go to the first maximum, ATen's rule)
    y2, idx = ops.maxpool3x3s2_idx(x.to(dev))
    assert torch.equal(y2, y) and idx.dtype == torch.uint8 and int(idx.max()) <= 8
    dx2 = ops.maxpool3x3s2_bwd_idx(idx, dy.to(dev), (H, W))
    assert torch.equal(dx2, dx)


def test_resize_and_layout(dev):
    from hallucidet_amd import ops
    x = torch.rand(2, 3, 64, 80)
    y = ops.nchw_to_nhwc_resize(x.to(dev), 30, 30, 8)
    want = ok.nearest_resize_nchw(x, 30, 30)
    assert torch.equal(y[..., :3].cpu().float(), ok.nchw_to_nhwc(want).half().float())
    assert (y[..., 3:] == 0).all()
    # BASELINE geometry index map: 512x640 -> 300x300
    xi = torch.arange(512 * 640, dtype=torch.float32).view(1, 1, 512, 640) % 2048
    yi = ops.nchw_to_nhwc_resize(xi.to(dev), 300, 300, 8)
    assert torch.equal(yi[..., 0].cpu().float(), ok.nearest_resize_nchw(xi, 300, 300)[0, 0].half().float()[None])
    # backward = adjoint
    xv = x.clone().requires_grad_(True)
    dy = rnd(2, 30, 30, 8, seed=41)
    ok.nearest_resize_nchw(xv, 30, 30).backward(ok.nhwc_to_nchw(dy[..., :3].float()))
    dx = ops.nchw_to_nhwc_resize_bwd(dy.to(dev), 2, 3, 64, 80, gscale=0.5)
    assert torch.allclose(dx.cpu(), 0.5 * xv.grad, atol=1e-6)
    # identity-size conversion and back
    y2 = ops.nchw_to_nhwc_resize(x.to(dev), 64, 80, 8)
    back = ops.nhwc_to_nchw(y2, 3)
    assert torch.equal(back.cpu(), x.half().float())


def test_upsample_add_and_bwd(dev):
    from hallucidet_amd import ops
    a, b = rnd(2, 19, 19, 16, seed=51), rnd(2, 10, 10, 16, seed=52)
    y = ops.upsample_add(a.to(dev), b.to(dev))
    bn = ok.nhwc_to_nchw(b.float()).requires_grad_(True)
    up = torch.nn.functional.interpolate(bn, size=[19, 19], mode="nearest")
    want = ok.nhwc_to_nchw(a.float()) + up
    close(y, ok.nchw_to_nhwc(want).half())
    dy = rnd(2, 19, 19, 16, seed=53)
    up.backward(ok.nhwc_to_nchw(dy.float()))
    db = torch.zeros(2, 10, 10, 16, dtype=torch.float16, device=dev)
    ops.upsample_add_bwd(dy.to(dev), db, accumulate=False)
    close(db, ok.nchw_to_nhwc(bn.grad).half(), rtol=2e-3, atol=2e-3)


def test_upsample2_bwd_slice_add(dev):
    from hallucidet_amd import ops
    dyu = rnd(2, 8, 12, 48, seed=61)
    dx = torch.zeros(2, 4, 6, 32, dtype=torch.float16, device=dev)
    ops.upsample2_bwd(dyu.to(dev), dx, 0, accumulate=False)
    want = dyu[..., :32].float().view(2, 4, 2, 6, 2, 32).sum(dim=(2, 4))
    close(dx, want.half(), rtol=2e-3, atol=2e-3)
    sk = torch.ones(2, 8, 12, 16, dtype=torch.float16, device=dev)
    ops.slice_channels(dyu.to(dev), sk, 32, accumulate=True)
    close(sk, (dyu[..., 32:].float() + 1).half())
    s = ops.add_f16(dyu.to(dev), dyu.to(dev))
    close(s, (2 * dyu.float()).half())
    z = torch.relu(rnd(2, 8, 12, 48, seed=62))
    r = ops.relu_bwd(dyu.to(dev), z.half().to(dev))
    assert torch.equal(r.cpu(), torch.where(z.half() > 0, dyu, torch.zeros_like(dyu)))


def test_subsample2(dev):
    from hallucidet_amd import ops
    x = rnd(2, 10, 10, 16, seed=71)
    y = ops.subsample2(x.to(dev))
    assert torch.equal(y.cpu(), x[:, ::2, ::2])
    x = rnd(2, 5, 5, 16, seed=72)
    y = ops.subsample2(x.to(dev))
    assert torch.equal(y.cpu(), x[:, ::2, ::2])
    dx = torch.ones(2, 5, 5, 16, dtype=torch.float16, device=dev)
    ops.subsample2_bwd(y, dx, accumulate=True)
    want = torch.ones(2, 5, 5, 16)
    want[:, ::2, ::2] += x[:, ::2, ::2].float()
    close(dx, want.half())


def test_sigmoid_bwd_and_channel_sum(dev):
    from hallucidet_amd import ops
    s = torch.rand(2, 3, 8, 16)
    dy = torch.randn(2, 3, 8, 16)
    dl = ops.sigmoid_bwd_nchw_to_nhwc(dy.to(dev), s.to(dev), 8, gscale=4.0)
    want = (4.0 * dy * s * (1 - s)).permute(0, 2, 3, 1)
    close(dl[..., :3], want.half())
    assert (dl[..., 3:] == 0).all()
    cs = ops.channel_sum(dl)
    assert torch.allclose(cs.cpu()[:3], dl.float().cpu().sum(dim=(0, 1, 2))[:3], rtol=1e-4, atol=1e-3)
    t = torch.randn(1000)
    assert torch.equal(ops.f16_to_f32(ops.f32_to_f16(t.to(dev), 2.0), 0.5).cpu(), (t * 2).half().float() * 0.5)
    out = torch.ones(8, device=dev)
    ops.scale_store(torch.arange(8.0, device=dev), out, 2.0, accumulate=True)
    assert torch.equal(out.cpu(), 1 + 2 * torch.arange(8.0))


def _random_boxes(n, seed, size=300.0):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * size * 0.8
    wh = torch.rand(n, 2, generator=g) * size * 0.3 + 2.0
    return torch.cat([xy, xy + wh], dim=1)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 700, 3375])
def test_nms_bit_exact(dev, n):
    from hallucidet_amd import ops
    B = 3
    nmax = n + 5
    boxes = torch.zeros(B, nmax, 4)
    counts = torch.tensor([n, max(1, n // 2), n], dtype=torch.int32)
    for b in range(B):
        bb = _random_boxes(int(counts[b]), seed=100 + b)
        if b == 2 and n > 4:
            bb[1::3] = bb[0:-1:3][: bb[1::3].shape[0]]  # exact duplicates (IoU == 1) and clustered boxes
        boxes[b, : counts[b]] = bb
    for thr in (0.5, 0.7):
        keep = ops.nms_sorted_batched(boxes.to(dev), counts.to(dev), thr).cpu()
        for b in range(B):
            c = int(counts[b])
            want = ok.nms_sorted(boxes[b, :c], thr)
            assert torch.equal(keep[b, :c], want), "NMS keep mask differs (n=%d thr=%.1f image %d)" % (n, thr, b)
            assert not keep[b, c:].any()


def test_roi_align_fwd_bwd(dev):
    from hallucidet_amd import ops
    N, H, W, C = 2, 19, 19, 16
    feat = rnd(N, H, W, C, seed=81)
    rois = torch.tensor([[0, 10.0, 20.0, 110.0, 220.0], [1, 0.0, 0.0, 299.0, 299.0], [1, 150.3, 40.2, 160.9, 47.7],
                         [0, 280.0, 280.0, 330.0, 310.0], [0, -20.0, -5.0, 30.0, 60.0]])
    scale = 1.0 / 16
    out = ops.roi_align(feat.to(dev), rois.to(dev), 7, 7, scale, 2)
    want = ok.roi_align_nchw(ok.nhwc_to_nchw(feat.float()), rois, 7, 7, scale, 2)
    close(out, want.permute(0, 2, 3, 1).half(), rtol=2e-3, atol=1e-3)
    # backward is the adjoint of the forward: <dout, J f> == <J^T dout, f>
    dout = rnd(5, 7, 7, C, seed=82)
    dfeat = ops.roi_align_bwd(dout.to(dev), rois.to(dev), (N, H, W, C), scale, 2).cpu()
    f2 = rnd(N, H, W, C, seed=83)
    out2 = ops.roi_align(f2.to(dev), rois.to(dev), 7, 7, scale, 2).float().cpu()
    lhs = (dout.float() * out2).sum()
    rhs = (dfeat * f2.float()).sum()
    assert abs(lhs - rhs) <= 2e-2 * max(1.0, abs(lhs)), (lhs, rhs)


def test_box_iou(dev):
    from hallucidet_amd import ops
    a, b = _random_boxes(7, 1), _random_boxes(1000, 2)
    got = ops.box_iou(a.to(dev), b.to(dev)).cpu()
    assert torch.equal(got, ok.box_iou(a, b))


def test_adam_step(dev):
    from hallucidet_amd import ops
    n = 10007
    g0 = torch.Generator().manual_seed(5)
    p, g, m, v = torch.randn(n, generator=g0), torch.randn(n, generator=g0) * 8, torch.randn(n, generator=g0) * 0.1, torch.rand(n, generator=g0)
    kw = dict(lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, clip_value=0.5, inv_scale=1.0 / 8, step=3)
    wp, wm, wv = ok.adam_reference(p, g, m, v, **kw)
    P, G, M, V = [t.clone().to(dev) for t in (p, g, m, v)]
    found = torch.zeros(1, device=dev)
    ops.adam_step(P, G, M, V, weight_decay=0.0, found_inf=found, **kw)
    assert torch.allclose(P.cpu(), wp, rtol=1e-6, atol=1e-7) and torch.allclose(M.cpu(), wm, rtol=1e-6, atol=1e-7) and torch.allclose(V.cpu(), wv, rtol=1e-6)
    # found_inf skips the step
    G[5] = float("inf")
    ops.check_finite(G, found)
    before = P.clone()
    ops.adam_step(P, G, M, V, weight_decay=0.0, found_inf=found, **kw)
    assert float(found) == 1.0 and torch.equal(P, before)


def test_inf_gradient_halves_the_scale_and_leaves_parameters_and_step_count(dev):
    """the scaler must act on the DEVICE found_inf flag: an overflowing step is skipped on the device, the scale
    halves before the next backward, Adam's step count (bias correction) does not advance, and a clean step then updates."""
    from hallucidet_amd.optim import FusedAdam, LossScaler, ParamArena
    ps = [torch.nn.Parameter(torch.randn(1000, device=dev)), torch.nn.Parameter(torch.randn(37, 5, device=dev))]
    arena = ParamArena(ps)
    opt = FusedAdam(arena, lr=1e-2, clip_value=0.5)
    sc = LossScaler(arena, init_scale=2.0 ** 16)
    loss = torch.tensor(1.0, device=dev)
    assert float(sc.scale(loss)) == 65536.0
    arena.flat_grads.fill_(0.25)
    arena.flat_grads[7] = float("inf")
    before = arena.flat_params.clone()
    sc.step(opt)
    sc.update()
    assert float(sc.scale(loss)) == 32768.0 and sc.scale_value == 32768.0           # resolved at the next scale(), no explicit sync
    assert torch.equal(arena.flat_params, before) and opt.step_count == 0 and opt.skipped_steps == 1
    arena.flat_grads.fill_(0.25)
    sc.step(opt)
    sc.update()
    sc.resolve()
    assert sc.scale_value == 32768.0 and opt.step_count == 1 and opt.skipped_steps == 1
    assert not torch.equal(arena.flat_params, before)
    want = before - 1e-2 * 0.25 / (0.25 + 1e-8)           # first Adam step: m_hat / (sqrt(v_hat) + eps) = g / (|g| + eps)
    assert torch.allclose(arena.flat_params[:1000], want[:1000], rtol=1e-5, atol=1e-6)


def test_roi_align_multilevel_bwd_patch_and_direct_paths(dev):
    """Multi-level RoIAlign backward: small / medium RoIs go through the LDS-patch kernels, large ones through direct
    atomics; all must equal the autograd of the oracle's RoIAlign."""
    from hallucidet_amd import ops
    from oracle import detection as od
    g = torch.Generator().manual_seed(3)
    C, N = 64, 2
    shapes = [(N, 75, 75, C), (N, 38, 38, C)]
    scales = [0.25, 0.125]
    feats = [rnd(*s, seed=90 + i) for i, s in enumerate(shapes)]
    sizes = [6.0, 14.0, 30.0, 70.0, 150.0, 280.0]
    rois, levels = [], []
    for k, sz in enumerate(sizes * 3):
        x1, y1 = float(torch.rand(1, generator=g) * (299 - sz) * 0.9), float(torch.rand(1, generator=g) * (299 - sz) * 0.9)
        ar = 0.5 + float(torch.rand(1, generator=g))
        rois.append([k % N, x1, y1, min(x1 + sz * ar, 320.0), min(y1 + sz / ar, 310.0)])
        levels.append(k % 2)
    rois.append([0, -30.0, -10.0, 20.0, 40.0])
    levels.append(0)
    rois = torch.tensor(rois)
    levels_t = torch.tensor(levels, dtype=torch.int32)
    R = rois.shape[0]
    dout = rnd(R, 7, 7, C, seed=95)
    dfs = ops.roi_align_ml_bwd(dout.to(dev), rois.to(dev), levels_t.to(dev), shapes, scales, 2)
    for l in range(2):
        f = ok.nhwc_to_nchw(feats[l].float()).requires_grad_(True)
        idx = [i for i in range(R) if levels[i] == l]
        out = od.roi_align_autograd(f, rois[idx], 7, scales[l], 2)
        out.backward(dout[idx].float().permute(0, 3, 1, 2))
        want = ok.nchw_to_nhwc(f.grad)
        got = dfs[l].cpu()
        assert torch.allclose(got, want, rtol=1e-3, atol=2e-3), (l, float((got - want).abs().max()))
    # forward of the multi-level kernel against the oracle as well
    out = ops.roi_align_ml([f.to(dev) for f in feats], scales, rois.to(dev), levels_t.to(dev), 7, 7, 2).float().cpu()
    for l in range(2):
        idx = [i for i in range(R) if levels[i] == l]
        want = od.roi_align_autograd(ok.nhwc_to_nchw(feats[l].float()), rois[idx], 7, scales[l], 2).permute(0, 2, 3, 1)
        close(out[idx], want.half(), rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize("C", [64, 256])
def test_roi_align_multilevel_bwd_gather_form(dev, C):
    """Gather-form RoIAlign backward (no atomics): equals the oracle's autograd for tiny / large / out-of-image RoIs, with
    more RoIs than one LDS list batch (1024), RoIs in arbitrary image order, trailing images without RoIs left zero, and is
    bit-reproducible."""
    from hallucidet_amd import ops
    from oracle import detection as od
    g = torch.Generator().manual_seed(4)
    N = 3                                        # C = 64: generic kernel; C = 256: the tile-weight kernel of the detector
    shapes = [(N, 75, 75, C), (N, 38, 38, C), (N, 19, 19, C)]
    scales = [0.25, 0.125, 0.0625]
    sizes = [2.0, 6.0, 14.0, 30.0, 70.0, 150.0, 280.0]
    rois, levels = [], []
    for k in range(1100):
        sz = sizes[k % len(sizes)]
        x1, y1 = float(torch.rand(1, generator=g) * (299 - sz) * 0.9), float(torch.rand(1, generator=g) * (299 - sz) * 0.9)
        ar = 0.5 + float(torch.rand(1, generator=g))
        rois.append([(k * 7) % 2, x1, y1, min(x1 + sz * ar, 320.0), min(y1 + sz / ar, 310.0)])     # images 0,1 only; interleaved
        levels.append(k % 3)
    rois += [[0, -30.0, -10.0, 20.0, 40.0], [1, 290.0, 295.0, 330.0, 340.0], [0, -50.0, -50.0, -20.0, -30.0], [1, 0.0, 0.0, 300.0, 300.0]]
    levels += [0, 1, 0, 2]
    rois = torch.tensor(rois)
    levels_t = torch.tensor(levels, dtype=torch.int32)
    R = rois.shape[0]
    dout = rnd(R, 7, 7, C, seed=96)
    dfs = ops.roi_align_ml_bwd_gather(dout.to(dev), rois.to(dev), levels_t.to(dev), shapes, scales, 2, n_images=2)
    dfs2 = ops.roi_align_ml_bwd_gather(dout.to(dev), rois.to(dev), levels_t.to(dev), shapes, scales, 2)
    for l in range(3):
        f = torch.zeros(N, C, shapes[l][1], shapes[l][2], requires_grad=True)
        idx = [i for i in range(R) if levels[i] == l]
        out = od.roi_align_autograd(f, rois[idx], 7, scales[l], 2)
        out.backward(dout[idx].float().permute(0, 3, 1, 2))
        want = ok.nchw_to_nhwc(f.grad)
        got = dfs[l].float().cpu()
        assert dfs[l].dtype == torch.float16 and got.shape == want.shape
        assert float(want[:2].abs().max()) > 1.0 and float(got[2].abs().max()) == 0.0
        err = (got - want).abs()
        assert float((err / (1e-2 + 2e-3 * want.abs())).max()) < 1.0, (l, float(err.max()), float(want.abs().max()))
        assert torch.equal(dfs[l], dfs2[l])                                   # deterministic; full-N form agrees
    # against the atomic form on the same inputs
    dfa = ops.roi_align_ml_bwd(dout.to(dev), rois.to(dev), levels_t.to(dev), shapes, scales, 2)
    for l in range(3):
        assert torch.allclose(dfs[l].float(), dfa[l], rtol=2e-3, atol=1e-2)
    with pytest.raises(Exception, match="7x7"):
        ops.roi_align_ml_bwd_gather(rnd(2, 5, 5, C, seed=1).to(dev), rois[:2].to(dev), levels_t[:2].to(dev), shapes, scales, 2)


def test_nms_topk_early_stop_equals_full_prefix(dev):
    """hd_nms_sorted_batched_topk: same keep flags as the full scan up to (and including) the 64-box chunk where the
    max_keep-th survivor falls, nothing kept after it; the first max_keep survivors are therefore identical."""
    from hallucidet_amd import ops
    g = torch.Generator().manual_seed(12)
    B, n = 3, 1500
    xy = torch.rand(B, n, 2, generator=g) * 200
    wh = torch.rand(B, n, 2, generator=g) * 60 + 4
    boxes = torch.cat([xy, xy + wh], dim=2).to(dev)
    counts = torch.tensor([1500, 900, 0], dtype=torch.int32, device=dev)
    full = ops.nms_sorted_batched(boxes, counts, 0.5)
    for mk in (1, 50, 300, 10 ** 6):
        part = ops.nms_sorted_batched(boxes, counts, 0.5, max_keep=mk)
        for b in range(B):
            kf, kp = full[b].cpu(), part[b].cpu()
            surv_f = torch.nonzero(kf).flatten()[:mk]
            surv_p = torch.nonzero(kp).flatten()[:mk]
            assert torch.equal(surv_f, surv_p), (mk, b)
            assert not (kp & ~kf).any()                                  # never keeps something the full scan suppresses
            if int(kf.sum()) > mk:
                last = int(surv_f[-1]) // 64 * 64 + 64
                assert torch.equal(kp[:last], kf[:last]) and not kp[last:].any()


@pytest.mark.gpu
@pytest.mark.parametrize("shared,allow_lq", [(True, True), (False, False), (False, True)])
def test_match_targets_equals_per_op_path(dev, shared, allow_lq):
    """hd_match_targets (fused IoU + Matcher + labels + encode) against the per-op path it replaces (box_iou_batched +
    detection._match_batched + label `where`s + BoxCoder.encode_single): match indices and labels bit-exact -- including
    padded GT rows, a GT-less image, duplicated GT boxes (ties -> first maximal index) -- regression targets to 1 ulp-ish."""
    from hallucidet_amd import ops
    import hallucidet_amd.models.detection as D
    torch.manual_seed(3)
    N, G, A = 5, 6, 3000
    xy = torch.rand(N, G, 2, device=dev) * 200
    gt = torch.cat([xy, xy + 8 + torch.rand(N, G, 2, device=dev) * 120], dim=2)
    gt[1, 3] = gt[1, 1]                                   # duplicate GT: tie between indices 1 and 3
    gvalid = torch.ones(N, G, dtype=torch.bool, device=dev)
    gvalid[0, 4:] = False
    gvalid[2, :] = False                                  # image without GT
    gt = torch.where(gvalid[:, :, None], gt, torch.zeros_like(gt))
    glabels = torch.randint(1, 5, (N, G), device=dev)
    bxy = torch.rand(N, A, 2, device=dev) * 260
    boxes = torch.cat([bxy, bxy + 4 + torch.rand(N, A, 2, device=dev) * 100], dim=2)
    boxes[:, :G] = gt                                     # exact overlaps (IoU 1) and equal-IoU ties
    if shared:
        boxes = boxes[0].contiguous()
    hi, lo = (0.7, 0.3) if allow_lq else (0.5, 0.5)
    w = (1.0, 1.0, 1.0, 1.0) if allow_lq else (10.0, 10.0, 5.0, 5.0)
    m, lab, reg = ops.match_targets(gt, gvalid, glabels, boxes, hi, lo, allow_lq, coder_weights=w)
    iou = ops.box_iou_batched(gt, boxes)
    m_ref = D._match_batched(iou, gvalid, hi, lo, allow_lq)
    assert torch.equal(m, m_ref)
    lab_ref = torch.gather(glabels, 1, m_ref.clamp(min=0))
    lab_ref = torch.where(m_ref == -1, torch.zeros_like(lab_ref), lab_ref)
    lab_ref = torch.where(m_ref == -2, torch.full_like(lab_ref, -1), lab_ref)
    lab_ref = torch.where(gvalid.any(dim=1)[:, None], lab_ref, torch.zeros_like(lab_ref))
    assert torch.equal(lab, lab_ref)
    m1, lab1, _ = ops.match_targets(gt, gvalid, None, boxes, hi, lo, allow_lq)
    assert torch.equal(m1, m_ref) and torch.equal(lab1 > 0, lab_ref > 0) and int(lab1.max()) <= 1
    matched = torch.gather(gt, 1, m_ref.clamp(min=0)[:, :, None].expand(-1, -1, 4)).reshape(-1, 4)
    props = (boxes[None].expand(N, -1, -1) if shared else boxes).reshape(-1, 4)
    reg_ref = D.BoxCoder(w).encode_single(matched, props).reshape(N, A, 4)
    pos = (m_ref >= 0) & gvalid.any(dim=1)[:, None]
    assert pos.any()
    assert torch.allclose(reg[pos], reg_ref[pos], rtol=1e-6, atol=1e-6)
    fin = torch.isfinite(reg_ref)
    assert torch.equal(torch.isfinite(reg), fin)


@pytest.mark.gpu
def test_topk_rows_segments_equals_stable_sort(dev):
    """hd_topk_select_rows against torch.sort(descending=True, stable=True)[1][:, :k] per segment: bit-exact index lists,
    including heavy ties (quantised scores), segments shorter than k, negative scores and -inf."""
    from hallucidet_amd import ops
    torch.manual_seed(11)
    segs = [16875, 4332, 1083, 300, 75]
    B, k = 6, 1000
    x = torch.randn(B, sum(segs), device=dev)
    x[1] = torch.round(x[1] * 4) / 4                       # many exact ties, also at the cut
    x[2] = -torch.rand(sum(segs), device=dev)              # all negative
    x[3, ::7] = float("-inf")
    x[4] = 0.0                                             # all equal: lowest indices win
    got = ops.topk_rows_segments(x, segs, k)
    ref, off = [], 0
    for n in segs:
        ref.append(torch.sort(x[:, off:off + n], dim=1, descending=True, stable=True)[1][:, :min(k, n)] + off)
        off += n
    assert torch.equal(got, torch.cat(ref, dim=1))
    got2 = ops.topk_rows_segments(x[:, :5000].contiguous(), [5000], 3000)     # P = 4096 path
    assert torch.equal(got2, torch.sort(x[:, :5000], dim=1, descending=True, stable=True)[1][:, :3000])


@pytest.mark.gpu
def test_fused_box_decoding_equals_per_op_path(dev):
    """hd_rpn_decode_filter / hd_roi_decode_clip against BoxCoder.decode_single + clip_boxes_to_image + the validity tests
    they replace: same boxes (<= 1 ulp of the exp), same validity flags."""
    from hallucidet_amd import ops
    import hallucidet_amd.models.detection as D
    torch.manual_seed(5)
    N, A, K = 3, 5000, 700
    axy = torch.rand(A, 2, device=dev) * 280
    anchors = torch.cat([axy, axy + 2 + torch.rand(A, 2, device=dev) * 150], dim=1)
    deltas = torch.randn(N, A, 4, device=dev) * 0.6
    deltas[0, :50, 2:] = 9.0                                # beyond bbox_xform_clip
    obj = torch.randn(N, A, device=dev) * 3
    top = torch.stack([torch.randperm(A, device=dev)[:K] for _ in range(N)])
    shape = (300, 300)
    coder = D.BoxCoder((1.0, 1.0, 1.0, 1.0))
    boxes, prob, valid = ops.rpn_decode_filter(deltas, obj, anchors, top, coder.bbox_xform_clip, shape, 1e-3, 0.0)
    ref_all = coder.decode_single(deltas.reshape(-1, 4), anchors.repeat(N, 1)).reshape(N, A, 4)
    bidx = torch.arange(N, device=dev)[:, None]
    ref = D.clip_boxes_to_image(ref_all[bidx, top], shape)
    assert torch.allclose(boxes, ref, rtol=2e-6, atol=1e-4)
    pref = torch.sigmoid(obj[bidx, top])
    assert torch.allclose(prob, pref, rtol=1e-6, atol=1e-7)
    vref = ((ref[..., 2] - ref[..., 0]) >= 1e-3) & ((ref[..., 3] - ref[..., 1]) >= 1e-3) & (pref >= 0.0)
    assert (valid != vref).float().mean() < 1e-3 and valid.any() and not valid.all()
    R, Kc = 900, 3
    rois = torch.cat([torch.randint(0, N, (R, 1), device=dev).float(), anchors[:R]], dim=1)
    codes = torch.randn(R, Kc * 4, device=dev)
    rc = D.BoxCoder((10.0, 10.0, 5.0, 5.0))
    got = ops.roi_decode_clip(codes, rois, rc.weights, rc.bbox_xform_clip, shape)
    want = D.clip_boxes_to_image(rc.decode_single(codes, rois[:, 1:]).reshape(R, Kc, 4), shape)
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-4)


@pytest.mark.gpu
def test_batched_nms_pick_equals_padded_path(dev):
    """hd_batched_nms_pick (shifted sorted boxes + mask + reduce emitting the ordered survivors) against the per-op path
    (_batched_nms_padded + prefix-sum compaction): identical survivor lists and counts, with invalid candidates, an image
    without any valid candidate and fewer survivors than top_n."""
    import hallucidet_amd.models.detection as D
    torch.manual_seed(9)
    B, n, top = 4, 1500, 300
    xy = torch.rand(B, n, 2, device=dev) * 250
    boxes = torch.cat([xy, xy + 5 + torch.rand(B, n, 2, device=dev) * 80], dim=2)
    scores = torch.rand(B, n, device=dev)
    scores[0, :200] = scores[0, 200:400]                  # exact score ties
    idxs = torch.randint(0, 4, (B, n), device=dev)
    valid = torch.rand(B, n, device=dev) > 0.2
    valid[2] = False
    valid[3, 40:] = False                                 # fewer survivors than top
    for thr in (0.7, 0.3):
        order, sel, counts = D._batched_nms_padded(boxes, scores, idxs, valid, thr, top)
        ref = torch.gather(order, 1, D._front(sel, top))
        pick, picked = D._batched_nms_pick(boxes, scores, idxs, valid, thr, top)
        assert torch.equal(picked, counts)
        assert torch.equal(pick, ref)
        assert int(picked[2]) == 0 and 0 < int(picked[3]) < top


@pytest.mark.gpu
def test_sample_pos_neg_equals_keyed_topk_path(dev):
    """hd_sample_pos_neg against the per-op keyed sampler (_sample_batched_keys) fed the SAME random keys: identical
    membership masks and counts -- more positives than the cap, fewer, none, and an image with nothing to sample."""
    from hallucidet_amd import ops
    import hallucidet_amd.models.detection as D
    torch.manual_seed(21)
    N, A, B, frac = 6, 9000, 256, 0.5
    labels = torch.randint(-1, 2, (N, A), device=dev, dtype=torch.int64)         # -1 / 0 / 1
    labels[1] = torch.where(torch.rand(A, device=dev) < 0.003, torch.ones_like(labels[1]), torch.zeros_like(labels[1]))   # few positives
    labels[2] = torch.clamp(labels[2], max=0)                                     # no positives
    labels[3] = -1                                                                # nothing to sample
    labels[4, 100:] = -1                                                          # fewer candidates than the batch
    keys = torch.randint(0, 1 << 30, (N, A), dtype=torch.int32, device=dev)
    pos_sel, neg_sel, counts = ops.sample_pos_neg(labels, keys, B, int(B * frac))
    pos, neg = labels >= 1, labels == 0
    big = torch.full_like(keys, 0x7FFFFFFF)
    k = min(B, A)
    kk = torch.stack([torch.where(pos, keys, big), torch.where(neg, keys, big)], dim=0).reshape(2 * N, A)
    order = torch.sort(kk, dim=1, stable=True)[1][:, :k].reshape(2, N, k)
    P, Nn = pos.sum(1), neg.sum(1)
    num_pos = P.clamp(max=int(B * frac))
    num_neg = torch.minimum(Nn, B - num_pos)
    ar = torch.arange(k, device=dev)[None, :]
    ref_p = torch.zeros_like(pos).scatter_(1, order[0], ar < num_pos[:, None])
    ref_n = torch.zeros_like(neg).scatter_(1, order[1], ar < num_neg[:, None])
    assert torch.equal(counts, torch.stack([num_pos, num_neg], dim=1))
    assert torch.equal(pos_sel, ref_p) and torch.equal(neg_sel, ref_n)
    assert int(counts[3].sum()) == 0 and int(counts[0, 0]) == int(B * frac) and int(counts[1, 0]) < int(B * frac)


C64_CASES = [
    # N, H, W, act, bias, res, mask   (3x3 / s1 / p1, 64 -> 64: conv3x3_c64.hip)
    (2, 16, 20, 0, False, False, False),      # fewer tiles than blocks
    (1, 13, 37, 1, True, False, False),       # ragged tiles in both directions, bias + ReLU (detector layer1)
    (3, 75, 75, 1, True, False, True),        # detector size, ReLU mask (data gradient), 150 tiles
    (2, 128, 160, 0, False, True, False),     # U-Net layer1 size: 640 tiles on 256 persistent blocks, residual (data gradient)
    (5, 64, 96, 0, False, False, False),      # 960 tiles: ragged tile runs per block
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", C64_CASES)
def test_conv_c64_register_resident_kernel(dev, case):
    """conv3x3_c64.hip (default route of the 64 -> 64 channel 3x3 layers) against the oracle and against the implicit-GEMM family
    (forced through the tuning override): same K order, same fp32 accumulation -> the outputs agree to an fp16 ulp; the BN partial rows
    differ in number (one per persistent block vs one per 128-row tile) but not in their totals."""
    from hallucidet_amd import ops, _abi
    N, H, W, act, use_bias, use_res, use_mask = case
    x = rnd(N, H, W, 64, seed=21)
    w = rnd(64, 576, scale=1.0 / 24.0, seed=22)
    bias = torch.randn(64, generator=torch.Generator().manual_seed(23)) if use_bias else None
    res = rnd(N, H, W, 64, seed=24) if use_res else None
    mask = (torch.rand(N, H, W, 64, generator=torch.Generator().manual_seed(25)) > 0.4).half() if use_mask else None
    d = lambda t: None if t is None else t.to(dev)
    lib = _abi.load()
    got, stats = ops.conv2d(d(x), d(w), 3, 3, bias=d(bias), res=d(res), mask=d(mask), pad=1, act=act, want_stats=True)
    tiles = N * ((H + 7) // 8) * ((W + 15) // 16)
    assert stats.shape[0] == min(tiles, 256)                       # the persistent kernel really ran
    lib.hd_conv_tune_override(128, 64, 64, 0)
    try:
        ref, rstats = ops.conv2d(d(x), d(w), 3, 3, bias=d(bias), res=d(res), mask=d(mask), pad=1, act=act, want_stats=True)
    finally:
        lib.hd_conv_tune_override(-1, -1, -1, -1)
    assert rstats.shape[0] == (N * H * W + 127) // 128
    torch.cuda.synchronize()
    assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * max(1.0, float(ref.float().abs().max()))
    assert torch.allclose(stats.sum(0), rstats.sum(0), rtol=2e-3, atol=2e-2)
    if N * H * W <= 20000:
        want, wstats = ok.conv2d_nhwc(x, w, 3, 3, bias=bias, res=res, pad=1, act=0)
        if use_mask:
            want = want * mask.float()
        if act == 1:
            want = want.clamp(min=0)
        close(got, want.half())
    # a second run is bit-identical (static tile runs, fixed summation order)
    got2, stats2 = ops.conv2d(d(x), d(w), 3, 3, bias=d(bias), res=d(res), mask=d(mask), pad=1, act=act, want_stats=True)
    assert torch.equal(got, got2) and torch.equal(stats, stats2)


SMALL_CASES = [c for c in CONV_CASES if c[6] == 3 and c[7] == 1 and c[8] == 1 and c[4] == 0 and c[3] in (8, 16, 32) and c[5] in (16, 32)
               and c[10] == 0 and not c[11] and not c[12]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", SMALL_CASES)
def test_conv_small_channel_kernel_equals_igemm(dev, case):
    """conv3x3_small.hip (default route of these shapes) against the implicit-GEMM family (forced through the tuning
    override): same fp32 accumulation of the same fp16 products, so outputs agree to the last fp16 bit or two; the BN
    partial-sum rows differ in number (8x32 tiles vs 128-row tiles) but not in their totals."""
    from hallucidet_amd import ops, _abi
    assert len(SMALL_CASES) >= 5
    N, H, W, C1, C2, Cout, K, stride, pad, up1, act, use_bias, use_res = case
    x = rnd(N, H, W, C1, seed=11).to(dev)
    w = rnd(Cout, 9 * C1, scale=1.0 / math.sqrt(9 * C1), seed=12).to(dev)
    lib = _abi.load()
    got, stats = ops.conv2d(x, w, 3, 3, pad=1, up1=up1, want_stats=True)
    Ho, Wo = got.shape[1], got.shape[2]
    assert stats.shape[0] == N * ((Ho + 7) // 8) * ((Wo + 31) // 32)
    lib.hd_conv_tune_override(128, 32, 32, 0)
    try:
        ref, rstats = ops.conv2d(x, w, 3, 3, pad=1, up1=up1, want_stats=True)
    finally:
        lib.hd_conv_tune_override(-1, -1, -1, -1)
    assert rstats.shape[0] == (N * Ho * Wo + 127) // 128          # the igemm family's 128-row tiles really ran
    assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * max(1.0, float(ref.float().abs().max()))
    assert torch.allclose(stats.sum(0), rstats.sum(0), rtol=2e-3, atol=2e-2)


@pytest.mark.gpu
def test_fused_detector_losses_equal_torch_ops(dev):
    """hd_rpn_loss / hd_fastrcnn_loss (+ backward) against the torch-op forms they replace: loss values to fp32 summation
    order, gradients element-wise."""
    import torch.nn.functional as F
    import hallucidet_amd.models.detection as D
    torch.manual_seed(8)
    T = 30000
    obj = (torch.randn(T, 1, device=dev) * 3).requires_grad_()
    dl = (torch.randn(T, 4, device=dev) * 0.3).requires_grad_()
    labels = (torch.rand(T, device=dev) < 0.3).float()
    reg_t = torch.randn(T, 4, device=dev) * 0.3
    samp = torch.rand(T, device=dev) < 0.05
    pos = samp & (labels > 0)
    for n_s in (torch.tensor(int(samp.sum()), device=dev), int(samp.sum()), 0):
        st = dict(labels=labels, reg_t=reg_t, pos_f=pos, samp_f=samp, n_sampled=n_s)
        lo, lb = D.rpn_loss_from_samples(st, obj, dl)
        go, gd = torch.autograd.grad(lo * 0.7 + lb * 1.3, (obj, dl))
        l1 = F.smooth_l1_loss(dl, torch.where(pos[:, None], reg_t, dl.detach()), beta=1 / 9, reduction="none").sum(dim=1)
        denom = n_s.clamp(min=1) if torch.is_tensor(n_s) else max(n_s, 1)
        rb = torch.where(pos, l1, torch.zeros_like(l1)).sum() / denom
        bce = F.binary_cross_entropy_with_logits(obj.flatten(), labels, reduction="none")
        ro = torch.where(samp, bce, torch.zeros_like(bce)).sum() / denom
        rgo, rgd = torch.autograd.grad(ro * 0.7 + rb * 1.3, (obj, dl))
        assert torch.allclose(lo, ro, rtol=2e-6, atol=1e-7) and torch.allclose(lb, rb, rtol=2e-6, atol=1e-7)
        assert torch.allclose(go, rgo, rtol=1e-5, atol=1e-9) and torch.allclose(gd, rgd, rtol=1e-5, atol=1e-9)
    R, K = 3000, 2
    lg = (torch.randn(R, K, device=dev) * 2).requires_grad_()
    br = (torch.randn(R, K * 4, device=dev) * 0.3).requires_grad_()
    lab = (torch.rand(R, device=dev) < 0.25).long()
    rt = torch.randn(R, 4, device=dev) * 0.3
    lc, lbx = D.fastrcnn_loss_flat(lg, br, lab, rt)
    g1 = torch.autograd.grad(lc * 0.4 + lbx * 1.1, (lg, br))
    rc = F.cross_entropy(lg, lab)
    brr = br.reshape(R, K, 4)
    picked = torch.gather(brr, 1, lab.clamp(min=0)[:, None, None].expand(-1, 1, 4)).squeeze(1)
    l1 = F.smooth_l1_loss(picked, torch.where((lab > 0)[:, None], rt, picked.detach()), beta=1 / 9, reduction="none").sum(dim=1)
    rbx = torch.where(lab > 0, l1, torch.zeros_like(l1)).sum() / R
    g2 = torch.autograd.grad(rc * 0.4 + rbx * 1.1, (lg, br))
    assert torch.allclose(lc, rc, rtol=2e-6, atol=1e-7) and torch.allclose(lbx, rbx, rtol=2e-6, atol=1e-7)
    assert torch.allclose(g1[0], g2[0], rtol=1e-5, atol=1e-9) and torch.allclose(g1[1], g2[1], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("cin,cout,up,head", [(32, 32, False, False), (32, 16, True, False), (16, 16, False, False), (16, 3, False, True), (8, 16, False, False)])
def test_consumer_side_batchnorm_is_bit_identical_to_bn_apply_then_conv(dev, cin, cout, up, head):
    """hd_conv2d / hd_wgrad with in_scale / in_shift (the producer's BatchNorm + ReLU folded into the operand staging of the
    small-channel 3x3 kernels, src/segmentation_models/base/modules.py:10-47) against hd_bn_apply followed by the plain call:
    outputs, BatchNorm partial sums and weight-gradient slabs must be IDENTICAL bit for bit -- ragged tile edges (the zero padding
    must stay zero after the affine map: shift != 0), the nearest-2x upsampled source and the fp32 NCHW head included."""
    from hallucidet_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + cout)
    N, H, W = 2, 21, 45                                  # not multiples of the 8 x 32 tile
    y_raw = (torch.randn(N, H, W, cin, generator=g) * 2).half().to(dev)
    scale = (torch.rand(cin, generator=g) + 0.5).to(dev)
    shift = (torch.randn(cin, generator=g) * 0.7 + 0.3).to(dev)          # relu(shift) != 0 at the padding if it were transformed
    cp = (cout + 7) // 8 * 8
    w = (torch.randn(cp, 9 * cin, generator=g) * 0.1).half().to(dev)
    z = ops.bn_apply(y_raw, scale, shift, relu=True)
    kw = dict(pad=1, up1=up)
    if head:
        bias = torch.randn(cout, generator=g).to(dev)
        kw.update(bias=bias, act=ops.ACT_SIGMOID, out_nchw_f32=True, cout=cout)
        a = ops.conv2d(z, w, 3, 3, **kw)
        b = ops.conv2d(y_raw, w, 3, 3, in_scale=scale, in_shift=shift, **kw)
        assert torch.equal(a, b)
    else:
        a, sa = ops.conv2d(z, w, 3, 3, want_stats=True, **kw)
        b, sb = ops.conv2d(y_raw, w, 3, 3, want_stats=True, in_scale=scale, in_shift=shift, **kw)
        assert torch.equal(a, b) and torch.equal(sa, sb)
        # no ReLU variant
        z2 = ops.bn_apply(y_raw, scale, shift, relu=False)
        assert torch.equal(ops.conv2d(z2, w, 3, 3, **kw), ops.conv2d(y_raw, w, 3, 3, in_scale=scale, in_shift=shift, in_relu=False, **kw))
    if cin >= 16:
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        dy = torch.randn(N, Ho, Wo, cp, generator=g).half().to(dev)
        s0 = ops.wgrad(z, dy, 3, 3, pad=1, up1=up, nsplit=7)
        s1 = ops.wgrad(y_raw, dy, 3, 3, pad=1, up1=up, nsplit=7, in_scale=scale, in_shift=shift)
        assert torch.equal(s0, s1) and float(s0.abs().sum()) > 0
    # shapes the small-channel kernels do not serve refuse the request instead of silently ignoring it
    x64 = torch.randn(1, 8, 8, 64, device=dev).half()
    w64 = torch.randn(64, 9 * 64, device=dev).half()
    with pytest.raises(Exception, match="consumer-side BatchNorm"):
        ops.conv2d(x64, w64, 3, 3, pad=1, in_scale=torch.ones(64, device=dev), in_shift=torch.zeros(64, device=dev))


def test_wgrad_reduce_multi_is_bit_identical_to_separate_reductions():
    """hd_wgrad_reduce_multi (every weight tensor of a backward segment in one launch) against hd_wgrad_reduce per tensor: all six
    reduction forms (plain, whole rows per block, one wave per quad, split x4 / x8 / x16), channel padding dropped, Cout < Cout_slab,
    accumulate."""
    from hallucidet_amd import ops
    torch.manual_seed(0)
    dev = "cuda"
    # (nsplit, Cout_slab, Cout, k, Cin, Cin_real, accumulate)
    shapes = [(3, 512, 512, 3, 512, 512, False),      # few slices, large tensor: row form in the batch (one 1 152-quad row per block, two trips), plain alone
              (16, 256, 256, 3, 256, 256, False),     # row form, one row of 576 quads per block
              (8, 208, 200, 3, 64, 64, True),         # row form, seven rows per block, ragged last block, Cout < Cout_slab, accumulate
              (5, 256, 256, 3, 768, 768, False),      # row form, 1 728 quads per row (decoder concat)
              (7, 48, 40, 3, 96, 96, False),          # row form, 216 quads per row, four rows per block
              (6, 96, 96, 1, 256, 256, False),        # 1x1: slab order == OIHW order -- stays on the plain form
              (768, 16, 16, 3, 16, 16, False),         # wave: 576 quads x 768 slices
              (256, 64, 64, 3, 64, 64, False),         # split x16 (9 216 quads)
              (64, 128, 128, 3, 128, 128, False),      # split x8 (36 864 quads)
              (16, 128, 128, 3, 64, 64, False),        # split x4
              (5, 64, 64, 7, 8, 3, False),             # stem: 3 real input channels of 8
              (256, 32, 24, 3, 128, 128, True),        # Cout < Cout_slab, accumulate
              (1, 8, 3, 3, 16, 16, False)]             # head: 3 of 8 rows, a single slice
    slabs, outs_a, outs_b = [], [], []
    batch = ops.WgradReduceBatch()
    for ns, cs, co, k, ci, cir, acc in shapes:
        slab = torch.randn(ns, cs, k * k * ci, device=dev)
        base = torch.randn(co, cir, k, k, device=dev)
        a, b = base.clone(), base.clone()
        ops.wgrad_reduce(slab, a, k, k, ci, Cin_real=cir, Cout=co, scale=0.37, accumulate=acc)
        batch.add(slab, b, k, k, ci, Cin_real=cir, Cout=co, scale=0.37, accumulate=acc)
        slabs.append(slab); outs_a.append(a); outs_b.append(b)
    batch.flush()
    torch.cuda.synchronize()
    for (ns, cs, co, k, ci, cir, acc), slab, a, b in zip(shapes, slabs, outs_a, outs_b):
        assert torch.equal(a, b), (ns, cs, co, k, ci)
        ref = slab.double().sum(0)[:co].reshape(co, k * k, ci)[:, :, :cir].permute(0, 2, 1).reshape(co, cir, k, k) * 0.37
        if not acc:
            assert torch.allclose(a.double(), ref, rtol=1e-4, atol=1e-3 * (ns ** 0.5))


def test_wgrad_multi_is_bit_identical_to_separate_launches_and_direct_output_to_the_reduction():
    """hd_wgrad_multi (the 8-wave 3x3 weight gradients of several layers as one grid) against hd_wgrad per layer with the same pixel
    split: slabs bit for bit, mixed shapes, a decoder concat, ragged maps, more entries than one grid holds (HD_WGRAD_MULTI_MAX = 24); a
    list with an entry the 8-wave kernel does not take falls back to separate launches.  hd_wgrad_args.dw_oihw (one pixel split: the
    kernel writes the scaled OIHW gradient itself) against hd_wgrad + hd_wgrad_reduce bit for bit, and against the fp32 definition."""
    from hallucidet_amd import ops
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(3)
    r = lambda *sh: (torch.randn(*sh, device=dev, generator=gen) * 0.5).half()
    # N, H, W, C1, C2 (skip of a concat; C1 then lives at half resolution), Cout, nsplit
    shapes = [(2, 16, 24, 64, 0, 64, 1), (2, 16, 24, 128, 0, 64, 3), (1, 19, 21, 64, 0, 128, 2), (2, 16, 16, 128, 64, 64, 1), (3, 8, 8, 256, 0, 256, 1),
              (1, 32, 40, 64, 0, 64, 7)] * 5                     # 30 entries: two grids
    calls, refs = [], []
    for N, H, W, C1, C2, Cout, ns in shapes:
        x = r(N, H // 2 if C2 else H, W // 2 if C2 else W, C1)
        x2 = r(N, H, W, C2) if C2 else None
        dy = r(N, H, W, Cout)
        kw = dict(x2=x2, pad=1, up1=bool(C2), nsplit=ns)
        assert ops.wgrad_takes_w8(x, dy, 3, 3, x2=x2, pad=1, up1=bool(C2))
        calls.append((x, dy, 3, 3, kw))
        refs.append(ops.wgrad(x, dy, 3, 3, **kw))
    slabs = ops.wgrad_multi(calls)
    torch.cuda.synchronize()
    for (sh, a, b) in zip(shapes, refs, slabs):
        assert a.shape == b.shape and torch.equal(a, b), sh
    # fallback: a 1x1 entry in the list
    x1, dy1 = r(2, 8, 8, 64), r(2, 8, 8, 64)
    mixed = calls[:3] + [(x1, dy1, 1, 1, dict(nsplit=2))]
    out = ops.wgrad_multi(mixed)
    torch.cuda.synchronize()
    assert torch.equal(out[3], ops.wgrad(x1, dy1, 1, 1, nsplit=2)) and all(torch.equal(a, b) for a, b in zip(out[:3], refs[:3]))
    # direct OIHW output
    calls_d, outs_d = [], []
    for (N, H, W, C1, C2, Cout, ns), (x, dy, _, _, kw), slab in zip(shapes[:6], calls[:6], refs[:6]):
        dw = torch.full((Cout, C1 + C2, 3, 3), 7.0, device=dev)
        calls_d.append((x, dy, 3, 3, dict(kw, dw=dw, dw_scale=0.37)))
        outs_d.append(dw)
    got = ops.wgrad_multi(calls_d)
    torch.cuda.synchronize()
    for (N, H, W, C1, C2, Cout, ns), (x, dy, _, _, kw), slab, dw, g in zip(shapes[:6], calls[:6], refs[:6], outs_d, got):
        want = torch.empty_like(dw)
        ops.wgrad_reduce(slab, want, 3, 3, C1 + C2, scale=0.37)
        if ns == 1:
            assert g is None and torch.equal(dw, want), (N, H, W, C1, C2, Cout)
            xin = x.float().permute(0, 3, 1, 2)
            if C2:
                xin = torch.cat([xin.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3), kw["x2"].float().permute(0, 3, 1, 2)], dim=1)
            ref = torch.nn.grad.conv2d_weight(xin, (Cout, C1 + C2, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1) * 0.37
            assert torch.allclose(dw, ref, rtol=2e-3, atol=2e-2), float((dw - ref).abs().max())
        else:                                  # more than one split: the slab comes back, dw is untouched
            assert g is not None and torch.equal(g, slab) and bool((dw == 7.0).all())
    with pytest.raises(Exception, match="dw_oihw"):
        from hallucidet_amd import _abi
        import ctypes as C
        a = _abi.WgradArgs(x1.data_ptr(), None, dy1.data_ptr(), None, 2, 8, 8, 8, 8, 64, 0, 8, 8, 64, 1, 1, 1, 0, 0, 1, None, None, 1, 0,
                           torch.empty(64, 64, 1, 1, device=dev).data_ptr(), 1.0, 0)
        ops.check(_abi.load().hd_wgrad(C.byref(a), None), "hd_wgrad")


def test_small_grid_conv_is_batch_invariant_and_run_to_run_identical(dev):
    """Image n of a batched launch equals the same image alone, and two runs agree, bit for bit, on a small-grid long-K layer (the
    property an in-launch split-K must keep; a ticketed split-K of the 64-deep family was built against this test, measured
    slower on every such layer of the step -- DESIGN 6 -- and removed)."""
    from hallucidet_amd import ops
    x = rnd(8, 10, 10, 512, seed=1).to(dev)
    w = rnd(512, 9 * 512, scale=1.0 / math.sqrt(9 * 512), seed=3).to(dev)
    m = (rnd(8, 10, 10, 512, seed=7) > 0).half().to(dev)
    from hallucidet_amd import _abi
    lib = _abi.load()
    a = ops.conv2d(x, w, 3, 3, pad=1, mask=m)
    b = ops.conv2d(x, w, 3, 3, pad=1, mask=m)
    # round 6: the tile model sees the launch's own batch -- differently batched launches agree to fp16 rounding ...
    one = ops.conv2d(x[5:6].contiguous(), w, 3, 3, pad=1, mask=m[5:6].contiguous())
    big = ops.conv2d(torch.cat([x, x, x]), w, 3, 3, pad=1, mask=torch.cat([m, m, m]))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    tol = 4e-3 * float(a.float().abs().max())
    assert float((a[5:6].float() - one.float()).abs().max()) <= tol and float((big[8:16].float() - a.float()).abs().max()) <= tol
    assert torch.equal(big[8:16], big[16:]) and torch.equal(big[:8], big[8:16]), "images of ONE launch: the same tile, the same bits"
    # ... and are bit-identical once the model is pinned to one batch (hd_conv_nominal_batch)
    lib.hd_conv_nominal_batch(8)
    try:
        a8 = ops.conv2d(x, w, 3, 3, pad=1, mask=m)
        one = ops.conv2d(x[5:6].contiguous(), w, 3, 3, pad=1, mask=m[5:6].contiguous())
        big = ops.conv2d(torch.cat([x, x, x]), w, 3, 3, pad=1, mask=torch.cat([m, m, m]))
    finally:
        lib.hd_conv_nominal_batch(0)
    assert torch.equal(a8[5:6], one)
    assert torch.equal(big[8:16], a8) and torch.equal(big[16:], a8)


@pytest.mark.parametrize("case", [(8, 16, 24, 128, 128), (3, 19, 21, 256, 256), (8, 16, 20, 512, 512), (2, 40, 24, 128, 256), (2, 24, 24, 64, 64)])
def test_fused_dgrad_wgrad_launch_is_bit_identical_to_the_two_launches(dev, case):
    """ops.wgrad_dgrad (hd_conv2d_wgrad: the data-gradient conv tiles and the 8-wave weight-gradient blocks of a layer as one grid)
    against ops.wgrad + ops.conv2d: same slab, same dx, bit for bit -- incl. a residual in the data-gradient epilogue, a shape whose
    convolution stays on the 4-wave family (64 -> 64: two launches inside the call) and ragged tiles."""
    from hallucidet_amd import ops
    N, H, W, Cin, Cout = case
    x = rnd(N, H, W, Cin, seed=1).to(dev)
    dy = rnd(N, H, W, Cout, seed=2).to(dev)
    wd = rnd(Cin, 9 * Cout, scale=1.0 / math.sqrt(9 * Cout), seed=3).to(dev)       # flipped / transposed layout: [Cin][tap][Cout]
    res = rnd(N, H, W, Cin, seed=4).to(dev)
    slab_a = ops.wgrad(x, dy, 3, 3, pad=1)
    dx_a = ops.conv2d(dy, wd, 3, 3, pad=1, cout=Cin, res=res)
    slab_b, dx_b = ops.wgrad_dgrad(x, dy, 3, 3, wd, pad=1, dgrad=dict(pad=1, cout=Cin, res=res))
    torch.cuda.synchronize()
    assert slab_a.shape == slab_b.shape and torch.equal(slab_a, slab_b)
    assert torch.equal(dx_a, dx_b)
    assert torch.isfinite(dx_b.float()).all() and float(slab_b.abs().max()) > 0


def test_conv2d_multi_equals_separate_calls(dev):
    """hd_conv2d_multi: the per-level convolutions of a detection head / FPN as one grid -- bit-identical to one hd_conv2d per level:
    a shared 3x3 conv over five feature levels (75^2 ... 5^2), its 1x1 heads with fp32 NHWC outputs and ragged channel counts,
    1x1 lateral convs with a different Cin per level, and data-gradient calls with residual + ReLU mask; plus a list the single
    grid cannot take (one member on the 8-wave family), which must fall back to separate launches with the same results."""
    from hallucidet_amd import ops
    g = lambda *s, seed: rnd(*s, seed=seed).to(dev)
    sizes = [(3, 75, 75), (3, 38, 38), (3, 19, 19), (3, 10, 10), (3, 5, 5)]
    feats = [g(n, h, w, 256, seed=10 + i) for i, (n, h, w) in enumerate(sizes)]
    w3 = g(256, 9 * 256, seed=1) * (1.0 / 48.0)
    bias = torch.randn(256, generator=torch.Generator().manual_seed(2)).to(dev)
    calls = [(f, w3, 3, 3, dict(bias=bias, pad=1, act=1)) for f in feats]
    sep = [ops.conv2d(x, w, kh, kw, **kw_) for x, w, kh, kw, kw_ in calls]
    got = ops.conv2d_multi(calls)
    for a, b in zip(sep, got):
        assert torch.equal(a, b)
    w1 = g(16, 256, seed=3) * (1.0 / 16.0)                         # 12 real output channels of 16 stored rows
    b1 = torch.randn(12, generator=torch.Generator().manual_seed(4)).to(dev)
    calls = [(t, w1, 1, 1, dict(bias=b1, out_nhwc_f32=True, cout=12)) for t in sep]
    for a, b in zip([ops.conv2d(x, w, kh, kw, **kw_) for x, w, kh, kw, kw_ in calls], ops.conv2d_multi(calls)):
        assert a.dtype == torch.float32 and torch.equal(a, b)
    cins = [256, 512, 1024, 2048]                                   # FPN lateral convs: another Cin on every level
    xs = [g(2, 40 >> i, 40 >> i, c, seed=20 + i) for i, c in enumerate(cins)]
    ws = [g(256, c, seed=30 + i) * (1.0 / math.sqrt(c)) for i, c in enumerate(cins)]
    calls = [(x, w, 1, 1, dict(bias=bias)) for x, w in zip(xs, ws)]
    for a, b in zip([ops.conv2d(x, w, kh, kw, **kw_) for x, w, kh, kw, kw_ in calls], ops.conv2d_multi(calls)):
        assert torch.equal(a, b)
    wd = g(256, 9 * 256, seed=5) * (1.0 / 48.0)                     # data gradients of the shared 3x3 conv, residual + ReLU mask
    dys = [g(*f.shape, seed=40 + i) for i, f in enumerate(feats)]
    masks = [(g(*f.shape, seed=50 + i) > 0).half() for i, f in enumerate(feats)]
    calls = [(d, wd, 3, 3, dict(pad=1, cout=256, res=f, mask=m, out_hw=(f.shape[1], f.shape[2]))) for d, f, m in zip(dys, feats, masks)]
    for a, b in zip([ops.conv2d(x, w, kh, kw, **kw_) for x, w, kh, kw, kw_ in calls], ops.conv2d_multi(calls)):
        assert torch.equal(a, b)
    big = g(8, 64, 80, 128, seed=60)                                # 128 -> 128 @64x80 runs in the 8-wave family: no single grid
    wb = g(128, 9 * 128, seed=61) * (1.0 / 34.0)
    calls = [(big, wb, 3, 3, dict(pad=1)), (g(2, 16, 16, 128, seed=62), wb, 3, 3, dict(pad=1))]
    for a, b in zip([ops.conv2d(x, w, kh, kw, **kw_) for x, w, kh, kw, kw_ in calls], ops.conv2d_multi(calls)):
        assert torch.equal(a, b)
    torch.cuda.synchronize()


@pytest.mark.parametrize("top_n", [1000, 150])
def test_batched_nms_over_segments_equals_the_one_list_form(dev, top_n):
    """hd_batched_nms_pick_segments (one greedy scan per (image, level), levels pre-sorted) against hd_batched_nms_pick (one scan per image
    over the globally sorted, category-shifted list): the same survivors in the same order, the same counts, the same padding -- with
    tied scores, invalid candidates inside the segments, an image whose candidates are all invalid and one truncated by top_n."""
    from hallucidet_amd.models import detection as D
    g = torch.Generator().manual_seed(5)
    seg = [1000, 1000, 700, 300, 75]
    B, n = 6, sum(seg)
    xy = torch.rand(B, n, 2, generator=g) * 260
    wh = torch.rand(B, n, 2, generator=g) * 90 + 4
    boxes = torch.cat([xy, xy + wh], dim=2)
    scores = torch.empty(B, n)
    lo = 0
    for s_ in seg:                                            # descending inside every segment, with runs of equal scores
        v = (torch.rand(B, s_, generator=g) * 64).floor() / 64
        scores[:, lo:lo + s_] = torch.sort(v, dim=1, descending=True)[0]
        lo += s_
    valid = torch.rand(B, n, generator=g) > 0.15
    valid[3] = False
    levels = torch.cat([torch.full((s_,), i, dtype=torch.int64) for i, s_ in enumerate(seg)])[None].expand(B, -1).contiguous()
    boxes, scores, valid, levels = boxes.to(dev), scores.to(dev), valid.to(dev), levels.to(dev)
    p0, c0 = D._batched_nms_pick(boxes, scores, levels, valid, 0.7, top_n)
    p1, c1 = D._batched_nms_pick_segments(boxes, scores, seg, valid, 0.7, top_n)
    torch.cuda.synchronize()
    assert torch.equal(c0.to(torch.int64), c1.to(torch.int64)), (c0, c1)
    assert int(c0[3]) == 0 and int(c0.max()) == min(top_n, int(c0.max()))
    assert torch.equal(p0, p1)


def test_random_shapes_against_aten_fp32_convolution(dev):
    """tools/fuzz_conv.py, a bounded slice: 60 shapes drawn from the dispatcher's whole domain (odd extents, ragged tiles, every kernel
    size / stride on the path, upsample + concat, every epilogue) -- forward, data gradient, weight gradient against ATen's fp32
    convolution on the GPU (not this repository's oracle), the fused and the multi-problem launches against their separate launches bit
    for bit.  The full sweep (8 300 legs, 0 failures) is profiles/r03_fuzz_conv.txt."""
    import importlib.util
    import os
    import random
    spec = importlib.util.spec_from_file_location("fuzz_conv", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_conv.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from hallucidet_amd import ops
    ran = 0
    for i in range(60):
        seed = 7_000_003 + i
        r = random.Random(seed)
        gen = torch.Generator(device="cuda").manual_seed(seed)
        c = fz.draw_case(r)
        legs = [fz.run_forward(ops, c, gen), fz.run_dgrad(ops, c, gen), fz.run_wgrad(ops, c, gen, r), fz.run_fused(ops, c, gen)]
        if i % 4 == 0:
            legs.append(fz.run_multi(ops, r, gen))
        if i % 4 == 1:
            legs.append(fz.run_consumer_bn(ops, r, gen))
        for msg in legs:
            if msg is None:
                continue
            ran += 1
            assert msg == "", (seed, c, msg)
    assert ran > 150


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_nms_on_a_coarse_grid_ties_at_the_threshold_and_degenerate_boxes(dev, seed):
    """NMS keep masks against the oracle, bit for bit, where the IoU test is decided by the last bit: boxes on a coarse integer or
    quarter-integer grid (IoU lands EXACTLY on 1/3, 1/2, 2/3 ... -- `>` vs `>=` and the order of the divide matter), many exact
    duplicates, zero-area and inverted boxes (IoU 0/0), very large coordinates; thresholds that are (0.5, 0.25) and are not (0.3, 0.7,
    1/3 as float) exactly representable."""
    from hallucidet_amd import ops
    g = torch.Generator().manual_seed(1000 + seed)
    B, n = 4, 333
    boxes = torch.zeros(B, n + 3, 4)
    # image 0: integer grid; 1: quarter grid; 2: integer grid shifted to 1e4 (large coordinates); 3: mixture with degenerate boxes
    for b in range(B):
        step = 1.0 if b != 1 else 0.25
        xy = torch.randint(0, 12, (n, 2), generator=g).float() * step
        wh = torch.randint(1, 7, (n, 2), generator=g).float() * step
        bb = torch.cat([xy, xy + wh], dim=1)
        if b == 2:
            bb = bb + 1.0e4
        if b == 3:
            bb[5::7, 2] = bb[5::7, 0]                    # zero width
            bb[6::11, 3] = bb[6::11, 1] - 1.0            # inverted in y
            bb[::13] = 0.0                               # the all-zero padding box
        boxes[b, :n] = bb
    counts = torch.tensor([n, n, n, n], dtype=torch.int32)
    for thr in (0.5, 0.25, 0.3, 0.7, float(torch.tensor(1.0 / 3.0))):
        keep = ops.nms_sorted_batched(boxes.to(dev), counts.to(dev), thr).cpu()
        for b in range(B):
            want = ok.nms_sorted(boxes[b, :n], thr)
            assert torch.equal(keep[b, :n].bool(), want.bool()), (seed, thr, b, int((keep[b, :n].bool() != want.bool()).sum()))
            assert not keep[b, n:].any()


def test_roi_align_kernels_reproduce_the_linear_ramp_closed_form(dev):
    """The analytic known answer of tests/test_third_party_pins.py held against the PRODUCT kernels (not the oracle): on a linear
    feature RoIAlign is the ramp at the bin centres.  fp32 storage (hd_roi_align_ml_f32: exact up to fp32 rounding) and fp16 storage
    (the fused-tap kernel of the detector's 7 x 7 / sampling-ratio-2 pooler: the ramp itself is rounded to fp16), two pyramid levels."""
    from hallucidet_amd import ops
    g = torch.Generator().manual_seed(11)
    a, b, c = 0.031, -0.017, 0.4
    feats32, scales, dims = [], [0.25, 0.125], [(40, 52), (20, 26)]
    for (H, W) in dims:
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
        ch = [(a * (k + 1)) * yy + (b * (k % 3 - 1)) * xx + c * k for k in range(16)]
        feats32.append(torch.stack(ch, dim=-1).float()[None].repeat(2, 1, 1, 1).contiguous())
    R = 40
    lv = torch.randint(0, 2, (R,), generator=g).to(torch.int32)
    rois = torch.zeros(R, 5)
    want = torch.zeros(R, 7, 7, 16, dtype=torch.float64)
    for r in range(R):
        H, W = dims[int(lv[r])]
        sc = scales[int(lv[r])]
        x0, y0 = float(torch.rand(1, generator=g)) * (W - 14) / sc, float(torch.rand(1, generator=g)) * (H - 14) / sc
        w, h = (float(torch.rand(1, generator=g)) * 11 + 0.3) / sc, (float(torch.rand(1, generator=g)) * 11 + 0.3) / sc
        rois[r] = torch.tensor([r % 2, x0, y0, x0 + w, y0 + h])
        sx0, sy0, sx1, sy1 = [float(torch.tensor(float(v) * sc, dtype=torch.float32)) for v in rois[r, 1:]]
        bw, bh = max(sx1 - sx0, 1.0) / 7, max(sy1 - sy0, 1.0) / 7
        for ph in range(7):
            for pw in range(7):
                cy, cx = sy0 + (ph + 0.5) * bh, sx0 + (pw + 0.5) * bw
                for k in range(16):
                    want[r, ph, pw, k] = (a * (k + 1)) * cy + (b * (k % 3 - 1)) * cx + c * k
    rois_d, lv_d = rois.to(dev), lv.to(dev)
    with ops.storage(torch.float32):
        got32 = ops.roi_align_ml([f.to(dev) for f in feats32], scales, rois_d, lv_d, 7, 7, 2)
    got16 = ops.roi_align_ml([f.half().to(dev) for f in feats32], scales, rois_d, lv_d, 7, 7, 2)
    torch.cuda.synchronize()
    assert float((got32.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float((got16.double().cpu() - want).abs().max()) <= 3e-3 * float(want.abs().max())


def test_roi_align_degenerate_and_outside_rois(dev):
    """RoIAlign(aligned=False, sampling_ratio 2) on the RoIs the reference's own path can produce at the edges: zero width / height
    (roi size clamps to 1), inverted, entirely outside the map on every side (samples beyond [-1, H] contribute zero), a single-pixel
    map corner, sub-pixel boxes and one covering 40x the map -- single-level and the multi-level (FPN) kernels, forward against the
    oracle and the backward as the forward's adjoint."""
    from hallucidet_amd import ops
    N, H, W, C = 2, 13, 17, 16
    feat = rnd(N, H, W, C, seed=91)
    s = 1.0 / 8
    rois = torch.tensor([[0, 40.0, 40.0, 40.0, 40.0],          # zero size
                         [1, 60.0, 30.0, 20.0, 10.0],          # inverted
                         [0, -400.0, -300.0, -200.0, -100.0],  # left / above
                         [1, 500.0, 400.0, 900.0, 700.0],      # right / below
                         [0, -8.0, -8.0, 0.0, 0.0],            # touches the corner pixel from outside: samples in [-1, 0]
                         [1, 135.9, 103.9, 136.1, 104.1],      # sub-pixel box at the far corner
                         [0, 33.3, 21.7, 33.9, 22.2],          # sub-pixel box inside
                         [1, -2000.0, -2000.0, 3000.0, 3000.0],  # 40x the map
                         [0, 0.0, 0.0, 135.99, 103.99]])       # the whole map
    out = ops.roi_align(feat.to(dev), rois.to(dev), 7, 7, s, 2)
    want = ok.roi_align_nchw(ok.nhwc_to_nchw(feat.float()), rois, 7, 7, s, 2)
    assert torch.isfinite(out).all()
    close(out, want.permute(0, 2, 3, 1).half(), rtol=2e-3, atol=1e-3)
    assert float(out[2].abs().max()) == 0.0 and float(out[3].abs().max()) == 0.0      # entirely outside: exactly zero
    dout = rnd(rois.shape[0], 7, 7, C, seed=92)
    dfeat = ops.roi_align_bwd(dout.to(dev), rois.to(dev), (N, H, W, C), s, 2).cpu()
    f2 = rnd(N, H, W, C, seed=93)
    out2 = ops.roi_align(f2.to(dev), rois.to(dev), 7, 7, s, 2).float().cpu()
    lhs, rhs = (dout.float() * out2).sum(), (dfeat * f2.float()).sum()
    assert torch.isfinite(dfeat).all() and abs(lhs - rhs) <= 2e-2 * max(1.0, abs(lhs)), (lhs, rhs)


def test_roi_align_one_block_per_roi_and_channel_lane_gather_at_256_channels(dev):
    """The detector's pooler at its channel count (round 4: hd_roi_align_ml runs one block per RoI, hd_roi_align_ml_bwd_gather lays the
    lanes of a wave along the channels; both accumulate with fused multiply-adds): forward against the oracle on regular, sub-pixel,
    zero-size, inverted, partly and entirely outside RoIs spread over three levels and two images, entirely-outside RoIs exactly zero;
    the backward is the forward's adjoint and leaves images without RoIs zero."""
    from hallucidet_amd import ops
    from oracle import detection as od
    C, N = 256, 3
    shapes = [(N, 40, 52, C), (N, 20, 26, C), (N, 10, 13, C)]
    scales = [0.25, 0.125, 0.0625]
    feats = [rnd(*s, seed=120 + i) for i, s in enumerate(shapes)]
    g = torch.Generator().manual_seed(12)
    rois, levels = [], []
    for k in range(90):
        sz = [3.0, 9.0, 27.0, 60.0, 150.0][k % 5]
        x1, y1 = float(torch.rand(1, generator=g) * 190), float(torch.rand(1, generator=g) * 150)
        ar = 0.5 + float(torch.rand(1, generator=g))
        rois.append([k % 2, x1, y1, x1 + sz * ar, y1 + sz / ar])
        levels.append(k % 3)
    special = [[0, 40.0, 40.0, 40.0, 40.0], [1, 60.0, 30.0, 20.0, 10.0], [0, -400.0, -300.0, -200.0, -100.0], [1, 500.0, 400.0, 900.0, 700.0],
               [0, -8.0, -8.0, 0.0, 0.0], [1, 207.9, 159.9, 208.1, 160.1], [0, 33.3, 21.7, 33.9, 22.2], [1, -2000.0, -2000.0, 3000.0, 3000.0],
               [0, 0.0, 0.0, 207.99, 159.99]]
    for k, r in enumerate(special):
        rois.append(r)
        levels.append(k % 3)
    rois = torch.tensor(rois)
    lv = torch.tensor(levels, dtype=torch.int32)
    R = rois.shape[0]
    out = ops.roi_align_ml([f.to(dev) for f in feats], scales, rois.to(dev), lv.to(dev), 7, 7, 2)
    assert out.dtype == torch.float16 and torch.isfinite(out).all()
    outc = out.float().cpu()
    for l in range(3):
        idx = [i for i in range(R) if levels[i] == l]
        want = od.roi_align_autograd(ok.nhwc_to_nchw(feats[l].float()), rois[idx], 7, scales[l], 2).permute(0, 2, 3, 1)
        close(outc[idx], want.half(), rtol=2e-3, atol=1e-3)
    assert float(outc[92].abs().max()) == 0.0 and float(outc[93].abs().max()) == 0.0          # entirely outside: exactly zero
    dout = rnd(R, 7, 7, C, seed=125)
    dfs = ops.roi_align_ml_bwd_gather(dout.to(dev), rois.to(dev), lv.to(dev), shapes, scales, 2, n_images=2)
    lhs = float((dout.float() * outc).sum())
    rhs = sum(float((d.float().cpu() * f.float()).sum()) for d, f in zip(dfs, feats))
    assert abs(lhs - rhs) <= 1e-2 * max(1.0, abs(lhs)), (lhs, rhs)
    assert all(float(d[2].abs().max()) == 0.0 for d in dfs)
    for l in range(3):
        f = torch.zeros(N, C, shapes[l][1], shapes[l][2], requires_grad=True)
        idx = [i for i in range(R) if levels[i] == l]
        o = od.roi_align_autograd(f, rois[idx], 7, scales[l], 2)
        o.backward(dout[idx].float().permute(0, 3, 1, 2))
        want = ok.nchw_to_nhwc(f.grad)
        err = (dfs[l].float().cpu() - want).abs()
        assert float((err / (1e-2 + 2e-3 * want.abs())).max()) < 1.0, (l, float(err.max()))


def test_fused_gradient_plumbing_equals_the_separate_passes(dev):
    """hd_concat_up_bwd == hd_upsample2_bwd + hd_slice_channels and hd_maxpool3x3s2_bwd_idx_add == hd_maxpool3x3s2_bwd_idx + hd_add_f16,
    bit for bit (round 4: four launches less per decoder block pair / stem in the U-Net's backward pass)."""
    from hallucidet_amd import ops
    for (N, H, W, cup, cskip) in [(2, 12, 20, 64, 32), (1, 6, 10, 32, 0), (3, 8, 8, 16, 8)]:
        dcat = rnd(N, H, W, cup + cskip, seed=130 + cup).to(dev)
        da, ds = ops.concat_up_bwd(dcat, cup)
        want_a = ops.upsample2_bwd(dcat, torch.empty(N, H // 2, W // 2, cup, device=dev, dtype=torch.float16), 0, accumulate=False)
        assert torch.equal(da, want_a)
        if cskip:
            want_s = ops.slice_channels(dcat, torch.empty(N, H, W, cskip, device=dev, dtype=torch.float16), cup, accumulate=False)
            assert torch.equal(ds, want_s)
        else:
            assert ds is None
    x = rnd(2, 17, 22, 16, seed=140).to(dev)
    y, idx = ops.maxpool3x3s2_idx(x)
    dy = rnd(*y.shape, seed=141).to(dev)
    other = rnd(*x.shape, seed=142).to(dev)
    got = ops.maxpool3x3s2_bwd_idx(idx, dy, (17, 22), add=other)
    want = ops.add_f16(ops.maxpool3x3s2_bwd_idx(idx, dy, (17, 22)), other)
    assert torch.equal(got, want)


@pytest.mark.parametrize("Cc", [256, 128, 64])            # 256 / 128: the 160- / 320-pixel tiles (conv3x3_m160.hip); 64: the register-resident 64 -> 64 kernel
@pytest.mark.parametrize("use_z,res", [(False, False), (True, True), (False, True)])
def test_batchnorm_backward_sums_from_the_data_gradient_epilogue(dev, use_z, res, Cc):
    """hd_conv_args.bs_*: the 8-wave 3x3 kernel that writes a unit's incoming gradient dz also emits the unit's BatchNorm backward sums.
    The summed rows equal hd_bn_bwd_reduce's on the stored dz (same expressions on the same fp16 values, different fp32 order), and the
    convolution output itself is bit-identical to the plain call."""
    from hallucidet_amd import ops
    N, H, W = (8, 32, 40) if Cc == 256 else ((8, 64, 80) if Cc == 128 else (3, 50, 70))          # (50 x 70: ragged 8 x 16 tiles)
    g = torch.Generator().manual_seed(77)
    dyv = (torch.randn(N, H, W, Cc, generator=g) * 0.3).half().to(dev)
    wd = (torch.randn(Cc, 9 * Cc, generator=g) / 48.0).half().to(dev)
    y_u = torch.randn(N, H, W, Cc, generator=g).half().to(dev)
    z_u = torch.relu(torch.randn(N, H, W, Cc, generator=g)).half().to(dev) if use_z else None
    r = (torch.randn(N, H, W, Cc, generator=g) * 0.2).half().to(dev) if res else None
    mean, invstd = torch.randn(Cc, generator=g).mul(0.1).to(dev), (torch.rand(Cc, generator=g) + 0.5).to(dev)
    gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(dev), torch.randn(Cc, generator=g).mul(0.2).to(dev)
    bs = dict(y=y_u, z=z_u, mean=mean, invstd=invstd, gamma=gamma, beta=beta, relu=True)
    dz = ops.conv2d(dyv, wd, 3, 3, pad=1, res=r, bstat=bs)
    assert bs["part"] is not None, "these 3x3 data gradients run in kernels that implement the sums"
    if Cc >= 128:
        assert bs["part"].shape[0] in (N * (H // 4) * (W // 40), N * (H // 8) * (W // 40)), "32x40x256 is the 160-pixel tile's shape (conv3x3_m160.hip, round 6)"
    plain = ops.conv2d(dyv, wd, 3, 3, pad=1, res=r)
    assert torch.equal(dz, plain)
    got = bs["part"].double().sum(0)
    rows = 64
    part = torch.empty(rows, 2 * Cc, device=dev)
    from hallucidet_amd import _abi
    ops.check(_abi.load().hd_bn_bwd_reduce(ops.ptr(dz), ops.ptr(z_u), ops.ptr(y_u), ops.ptr(mean), ops.ptr(invstd), ops.ptr(gamma), ops.ptr(beta),
                                           ops.ptr(part), rows, N * H * W, Cc, 1, ops._stream()), "hd_bn_bwd_reduce")
    want = part.double().sum(0)
    scale = want.abs().max()
    assert float((got - want).abs().max()) <= 2e-5 * float(scale), (float((got - want).abs().max()), float(scale))
    # and through bn_backward: same dy either way
    a = ops.bn_backward(dz, z_u, y_u, mean, invstd, gamma, beta, relu=True, part=bs["part"])
    b = ops.bn_backward(dz, z_u, y_u, mean, invstd, gamma, beta, relu=True)
    assert float((a[0].float() - b[0].float()).abs().max()) <= 2e-3 * float(b[0].float().abs().max())
    assert torch.allclose(a[2], b[2], rtol=1e-4, atol=1e-4 * float(b[2].abs().max())) and torch.allclose(a[3], b[3], rtol=1e-4, atol=1e-4 * float(b[3].abs().max()))


@pytest.mark.parametrize("cfg", [18, 19, 20])
def test_batchnorm_backward_sums_on_ragged_160_pixel_tiles(dev, cfg):
    """The bs_* epilogue on maps the 40-pixel-wide tiles cover raggedly (3 x 19 x 21 and 2 x 10 x 10 on 256 channels, residual + ReLU mask from
    z): a thread's rows sit in different COLUMNS there, so a row inside the map can follow one outside it (a bug of the first version:
    the y / z address of every row was clamped by the validity of the thread's first row).  Sums against hd_bn_bwd_reduce on the stored dz."""
    from hallucidet_amd import ops, _abi
    lib = _abi.load()
    Cc = 256
    for (N, H, W) in [(3, 19, 21), (2, 10, 10)]:
        g = torch.Generator().manual_seed(78)
        dyv = (torch.randn(N, H, W, Cc, generator=g) * 0.3).half().to(dev)
        wd = (torch.randn(Cc, 9 * Cc, generator=g) / 48.0).half().to(dev)
        y_u = torch.randn(N, H, W, Cc, generator=g).half().to(dev)
        z_u = torch.relu(torch.randn(N, H, W, Cc, generator=g)).half().to(dev)
        r = (torch.randn(N, H, W, Cc, generator=g) * 0.2).half().to(dev)
        mean, invstd = torch.randn(Cc, generator=g).mul(0.1).to(dev), (torch.rand(Cc, generator=g) + 0.5).to(dev)
        gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(dev), torch.randn(Cc, generator=g).mul(0.2).to(dev)
        bs = dict(y=y_u, z=z_u, mean=mean, invstd=invstd, gamma=gamma, beta=beta, relu=True)
        try:
            lib.hd_conv_tune_w8(cfg, 1)
            dz = ops.conv2d(dyv, wd, 3, 3, pad=1, res=r, bstat=bs)
        finally:
            lib.hd_conv_tune_w8(-1, 0)
        th, tw = (8 if cfg == 19 else 4), (24 if cfg == 20 else 40)
        assert bs["part"] is not None and bs["part"].shape[0] == N * ((H + th - 1) // th) * ((W + tw - 1) // tw)
        rows = 64
        part = torch.empty(rows, 2 * Cc, device=dev)
        ops.check(lib.hd_bn_bwd_reduce(ops.ptr(dz), ops.ptr(z_u), ops.ptr(y_u), ops.ptr(mean), ops.ptr(invstd), ops.ptr(gamma), ops.ptr(beta),
                                       ops.ptr(part), rows, N * H * W, Cc, 1, ops._stream()), "hd_bn_bwd_reduce")
        got, want = bs["part"].double().sum(0), part.double().sum(0)
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), (N, H, W, float((got - want).abs().max()), float(want.abs().max()))


@pytest.mark.parametrize("shape", [(2, 300, 300), (1, 75, 101), (3, 64, 38)])
def test_stem_data_gradient_in_sub_pixel_form(dev, shape):
    """hd_conv7x7s2_dgrad_thin (round 4) against the implicit-GEMM route it replaces (hd_conv2d with in_dil = 2 after hd_relu_bwd) and
    against ATen's fp32 transposed convolution: even and odd extents, ragged tiles, the ReLU mask of the stem fused into the staging."""
    from hallucidet_amd import ops
    N, H, W = shape
    Hl, Wl = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    g = torch.Generator().manual_seed(91)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    wf, wd = ops.weight_prep(w.to(dev), cin_pad=8, cout_pad=64, want_fwd=True, want_dgrad=True)
    dy = (torch.randn(N, Hl, Wl, 64, generator=g) * 0.5).half().to(dev)
    z = torch.relu(torch.randn(N, Hl, Wl, 64, generator=g)).half().to(dev)
    w16 = ops.stem_dgrad_weights(wf, 8)
    got = ops.conv7x7s2_dgrad_thin(dy, w16, (H, W), mask_z=z)
    masked = ops.relu_bwd(dy, z)
    old = ops.conv2d(masked, wd, 7, 7, stride=1, pad=3, in_dil=2, out_hw=(H, W), cout=8)
    assert got.shape == old.shape == (N, H, W, 8) and float(got[..., 3:].abs().max()) == 0.0
    err = float((got.float() - old.float()).abs().max())
    assert err <= 2e-3 * float(old.float().abs().max()) + 1e-3, err
    want = torch.nn.functional.conv_transpose2d(masked.float().permute(0, 3, 1, 2), wf.view(64, 7, 7, 8)[..., :3].float().permute(0, 3, 1, 2).contiguous(),
                                                stride=2, padding=3, output_padding=(H + 6 - 7 - 2 * (Hl - 1), W + 6 - 7 - 2 * (Wl - 1)))
    assert want.shape[2:] == (H, W)
    close(got[..., :3].cpu(), want.permute(0, 2, 3, 1).half().cpu(), rtol=3e-3, atol=3e-3)
    assert torch.equal(ops.conv7x7s2_dgrad_thin(dy, w16, (H, W)), ops.conv7x7s2_dgrad_thin(dy, w16, (H, W), mask_z=torch.ones_like(z)))


STEM_CASES = [
    # N, H, W, act, bias   (7x7 / s2 / p3, 8 -> 64: conv7x7s2_stem.hip)
    (2, 64, 80, 0, False),        # fewer tiles than blocks
    (1, 75, 101, 1, True),        # odd extents, ragged tiles, bias + ReLU (the detector's stem)
    (3, 300, 300, 1, True),       # detector size
    (2, 512, 640, 0, False),      # U-Net size: 1 280 tiles on 256 persistent blocks, BatchNorm partial sums
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", STEM_CASES)
def test_conv_stem_register_resident_kernel(dev, case):
    """conv7x7s2_stem.hip (default route of the ResNet stem) against ATen's fp32 convolution on the GPU and against the implicit-GEMM
    family (forced through the tuning override): outputs to an fp16 ulp, BatchNorm partial rows equal in their totals, runs bit-identical."""
    from hallucidet_amd import ops, _abi
    N, H, W, act, use_bias = case
    g = torch.Generator().manual_seed(31)
    x = torch.zeros(N, H, W, 8)
    x[..., :3] = torch.rand(N, H, W, 3, generator=g)
    x = x.half()
    w4 = torch.zeros(64, 7, 7, 8)
    w4[..., :3] = torch.randn(64, 7, 7, 3, generator=g) * 0.08
    w = w4.view(64, 392).half()
    bias = torch.randn(64, generator=g) * 0.3 if use_bias else None
    d = lambda t: None if t is None else t.to(dev)
    lib = _abi.load()
    got, stats = ops.conv2d(d(x), d(w), 7, 7, bias=d(bias), stride=2, pad=3, act=act, want_stats=True)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert got.shape == (N, Ho, Wo, 64)
    tiles = N * ((Ho + 7) // 8) * ((Wo + 15) // 16)
    assert stats.shape[0] == min(tiles, 256)                       # the persistent kernel really ran
    lib.hd_conv_tune_override(128, 64, 32, 0)
    try:
        ref, rstats = ops.conv2d(d(x), d(w), 7, 7, bias=d(bias), stride=2, pad=3, act=act, want_stats=True)
    finally:
        lib.hd_conv_tune_override(-1, -1, -1, -1)
    assert rstats.shape[0] != stats.shape[0] or tiles <= 256
    torch.cuda.synchronize()
    assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * max(1.0, float(ref.float().abs().max()))
    assert torch.allclose(stats.sum(0), rstats.sum(0), rtol=2e-3, atol=2e-2)
    want = torch.nn.functional.conv2d(d(x).float().permute(0, 3, 1, 2), d(w).float().view(64, 7, 7, 8).permute(0, 3, 1, 2), bias=d(bias), stride=2, padding=3)
    if act == 1:
        want = want.clamp(min=0)
    close(got.cpu(), want.permute(0, 2, 3, 1).half().cpu(), rtol=3e-3, atol=3e-3)
    got2, stats2 = ops.conv2d(d(x), d(w), 7, 7, bias=d(bias), stride=2, pad=3, act=act, want_stats=True)
    assert torch.equal(got, got2) and torch.equal(stats, stats2)
    # image n of the batch == the same image alone (tile choice and K order do not depend on N)
    one = ops.conv2d(d(x[:1]), d(w), 7, 7, bias=d(bias), stride=2, pad=3, act=act)
    assert torch.equal(one[0], got[0])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 16, 20), (1, 13, 37), (8, 256, 320)])
def test_conv_32_to_128_register_resident_kernel(dev, shape):
    """conv3x3_c32to128.hip (decoder block 3's data gradient: 32 -> 128 channels) against the implicit-GEMM family (forced through the
    tuning override; same K order: outputs to an fp16 ulp) and ATen's fp32 convolution; runs bit-identical, image n == the image alone."""
    from hallucidet_amd import ops, _abi
    N, H, W = shape
    x = rnd(N, H, W, 32, seed=41).to(dev)
    w = rnd(128, 288, scale=1.0 / 17.0, seed=42).to(dev)
    lib = _abi.load()
    got = ops.conv2d(x, w, 3, 3, pad=1)
    lib.hd_conv_tune_override(128, 32, 32, 0)
    try:
        ref = ops.conv2d(x, w, 3, 3, pad=1)
    finally:
        lib.hd_conv_tune_override(-1, -1, -1, -1)
    torch.cuda.synchronize()
    assert got.shape == (N, H, W, 128)
    assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * max(1.0, float(ref.float().abs().max()))
    want = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(128, 3, 3, 32).permute(0, 3, 1, 2), padding=1)
    close(got.cpu(), want.permute(0, 2, 3, 1).half().cpu(), rtol=3e-3, atol=3e-3)
    assert torch.equal(got, ops.conv2d(x, w, 3, 3, pad=1))
    assert torch.equal(ops.conv2d(x[:1].contiguous(), w, 3, 3, pad=1)[0], got[0])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 10), (1, 7, 19), (8, 128, 160)])
def test_conv_upsample_concat_128_to_32_register_resident_kernel(dev, shape):
    """conv3x3_cat128to32.hip (decoder block 3's first convolution: cat([nearest_2x(a), skip]) 64 + 64 -> 32 channels, with BatchNorm
    partial sums) against the implicit-GEMM family (tuning override; outputs to an fp16 ulp, sums equal in their totals) and ATen's fp32
    convolution of the materialised upsample + concat; runs bit-identical, image n == the image alone."""
    from hallucidet_amd import ops, _abi
    N, Hs, Ws = shape                                 # low-resolution extent; the output is 2 Hs x 2 Ws
    a = rnd(N, Hs, Ws, 64, seed=51).to(dev)
    skip = rnd(N, 2 * Hs, 2 * Ws, 64, seed=52).to(dev)
    w = rnd(32, 9 * 128, scale=1.0 / 34.0, seed=53).to(dev)
    lib = _abi.load()
    got, stats = ops.conv2d(a, w, 3, 3, x2=skip, up1=True, pad=1, want_stats=True)
    tiles = N * ((2 * Hs + 7) // 8) * ((2 * Ws + 15) // 16)
    assert got.shape == (N, 2 * Hs, 2 * Ws, 32) and stats.shape[0] == min(tiles, 256)          # the persistent kernel really ran
    lib.hd_conv_tune_override(128, 32, 64, 0)
    try:
        ref, rstats = ops.conv2d(a, w, 3, 3, x2=skip, up1=True, pad=1, want_stats=True)
    finally:
        lib.hd_conv_tune_override(-1, -1, -1, -1)
    torch.cuda.synchronize()
    assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * max(1.0, float(ref.float().abs().max()))
    assert torch.allclose(stats.sum(0), rstats.sum(0), rtol=2e-3, atol=2e-2)
    up = a.float().permute(0, 3, 1, 2).repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    cat = torch.cat([up, skip.float().permute(0, 3, 1, 2)], dim=1)
    want = torch.nn.functional.conv2d(cat, w.float().view(32, 3, 3, 128).permute(0, 3, 1, 2), padding=1)
    close(got.cpu(), want.permute(0, 2, 3, 1).half().cpu(), rtol=3e-3, atol=3e-3)
    got2, stats2 = ops.conv2d(a, w, 3, 3, x2=skip, up1=True, pad=1, want_stats=True)
    assert torch.equal(got, got2) and torch.equal(stats, stats2)
    assert torch.equal(ops.conv2d(a[:1].contiguous(), w, 3, 3, x2=skip[:1].contiguous(), up1=True, pad=1)[0], got[0])
