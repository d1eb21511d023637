"""Hallucination network: U-Net with a ResNet encoder, executed by hand-written HIP kernels.

Host-side mirror of the reference's `smp.Unet` for the one configuration the hot path uses
(src/models/encoder_decoder.py:22-30; src/segmentation_models/decoders/unet/model.py:56-100,
decoders/unet/decoder.py:11-46,68-124; base/modules.py:10-47; base/heads.py:21-27; encoders/resnet.py:37-65).
Sub-module names and parameter shapes equal the reference's checkpoint layout:
  encoder.{conv1,bn1,layer1..4.{i}.{conv1,bn1,conv2,bn2,downsample.{0,1}}},
  decoder.blocks.{i}.conv{1,2}.{0,1}, segmentation_head.{0,1,2}.

torch.nn.Conv2d / BatchNorm2d objects are used ONLY as parameter containers (names, shapes, state_dict,
initialisation); their forward() is never called.  `Unet.forward` runs an explicit schedule of C-ABI launches
(`UnetRunner`) and the whole forward/backward is exposed to autograd as ONE Function, so PyTorch owns memory and the
graph edge while every FLOP runs in libhallucidet_hip.so.  There is no ATen fallback.

Data layout in HBM: activations NHWC fp16; per conv a raw output `y` and a post-BN/ReLU tensor `z` are kept for the
backward pass; weights are re-packed each step from the fp32 masters into [Cout][KH][KW][Cin] fp16 (+ the flipped
[Cin][KH][KW][Cout] copy the data-gradient consumes); BatchNorm statistics, parameter gradients and the optimizer run
in fp32 on one flat arena (one all-reduce buffer for data parallelism).
"""
import contextlib
import os

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_NONE, ACT_SIGMOID


# ----------------------------------------------------------------------------------------------------------------
# parameter containers (structure only)
# ----------------------------------------------------------------------------------------------------------------

@contextlib.contextmanager
def capture_without_gc():
    """Python's cyclic collector can run at any allocation -- also inside a stream capture, where collecting an unreachable
    CUDAGraph or a tensor of a captured pool (a graph dropped by a re-capture, an earlier module) calls hipGraphDestroy / hipFree
    while the stream is capturing, and the runtime aborts the process (seen once in the test suite: 'Fatal Python error: Aborted',
    'Garbage-collecting' on top of the stack, inside the forward capture).  Collect first, keep the collector off while capturing."""
    import gc
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride


class Bottleneck(nn.Module):
    """torchvision Bottleneck [EXT] (v1.5: the stride sits on the 3x3 conv), the block of the reference's resnet50 encoder
    (src/segmentation_models/encoders/resnet.py:136-144)."""
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        cout = planes * self.expansion
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride


class ResNetEncoder(nn.Module):
    """torchvision ResNet(BasicBlock | Bottleneck, layers) minus avgpool/fc, as the reference's ResNetEncoder exposes it."""

    def __init__(self, layers=(3, 4, 6, 3), out_channels=(3, 64, 64, 128, 256, 512), depth=5, block="basic"):
        super().__init__()
        Block = Bottleneck if block == "bottleneck" else BasicBlock
        self._depth, self._out_channels, self._in_channels = depth, out_channels, 3
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (n, c) in enumerate(zip(layers, (64, 128, 256, 512))):
            blocks = []
            for b in range(n):
                blocks.append(Block(cin, c, 2 if (b == 0 and i > 0) else 1))
                cin = c * Block.expansion
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    @property
    def out_channels(self):
        return self._out_channels[: self._depth + 1]

    @property
    def output_stride(self):
        return min(32, 2 ** self._depth)


class Conv2dReLU(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, padding=1, stride=1, use_batchnorm=True):
        if not use_batchnorm or use_batchnorm == "inplace":
            raise NotImplementedError("hallucidet_amd: only use_batchnorm=True is on the hot path")
        super().__init__(nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, skip_channels, out_channels, use_batchnorm=True, attention_type=None):
        super().__init__()
        if attention_type is not None:
            raise NotImplementedError("hallucidet_amd: decoder_attention_type must be None (reference default)")
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels, use_batchnorm=use_batchnorm)
        self.attention1 = nn.Identity()
        self.conv2 = Conv2dReLU(out_channels, out_channels, use_batchnorm=use_batchnorm)
        self.attention2 = nn.Identity()
        self.in_channels, self.skip_channels = in_channels, skip_channels


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels, n_blocks=5, use_batchnorm=True, attention_type=None, center=False):
        super().__init__()
        if n_blocks != len(decoder_channels):
            raise ValueError("Model depth is {}, but you provide `decoder_channels` for {} blocks.".format(n_blocks, len(decoder_channels)))
        if center:
            raise NotImplementedError("hallucidet_amd: center block is only used by vgg encoders (out of scope)")
        enc = list(encoder_channels[1:])[::-1]
        cins = [enc[0]] + list(decoder_channels[:-1])
        cskips = enc[1:] + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList(DecoderBlock(a, b, c, use_batchnorm, attention_type) for a, b, c in zip(cins, cskips, decoder_channels))


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, activation=None, upsampling=1):
        if upsampling != 1:
            raise NotImplementedError("hallucidet_amd: head upsampling is not on the hot path")
        act = nn.Identity() if activation is None else activation
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, padding=kernel_size // 2), nn.Identity(), act)


def initialize_decoder(module):
    """base/initialization.py:4-19"""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


def initialize_head(module):
    """base/initialization.py:22-27"""
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


_WRED_MULTI = os.environ.get("HD_WRED_MULTI", "1") != "0"     # A/B knob: one slab-reduction launch per backward segment
_WGRAD_DEFER = os.environ.get("HD_WGRAD_DEFER", "1") != "0"   # A/B knob: a backward segment's 8-wave weight gradients as one grid at its end
_EXCHANGE_BUCKETS = 2 if os.environ.get("HD_EXCHANGE_BUCKETS", "2") != "5" else 5     # gradient-exchange buckets / backward graphs with a hook: 2 (default) or 5 (one per segment)
_WGRAD_MERGE = os.environ.get("HD_WGRAD_MERGE", "1") != "0"     # A/B knob: without an exchange hook, launch the deferred weight gradients of segments 0-2 / 3-4 together
_WGRAD_DIRECT = os.environ.get("HD_WGRAD_DIRECT", "1") != "0"   # A/B knob: one-split weight gradients written as the OIHW gradient by the kernel
_WGRAD_DEFER_BLOCKS = int(os.environ.get("HD_WGRAD_DEFER_BLOCKS", "0"))     # > 0: force that grid size (A/B); 0: simulated schedule
# A/B knob: without an exchange hook the two deferred weight-gradient grids run on a SIDE stream (a parallel branch of the backward graph):
# group 1 (decoder, layer4, layer3) beside the BatchNorm-backward / data-gradient chain of layer2 and layer1, group 2 beside the max-pool
# and stem backward.  Same kernels, same operands, same results -- only the order in which the chip sees them changes.
_WGRAD_SIDE = int(os.environ.get("HD_WGRAD_SIDE", "0"))      # 1: both groups, 2: group 1 only, 3: group 2 only
_POOL2 = os.environ.get("HD_POOL2", "1") != "0"               # A/B knob: 2x2 sum-pool of the last decoder block's data gradient in its epilogue

def _plan_wgrad_splits(geo, cus=256):
    """Pixel splits of the layers of one multi-layer weight-gradient grid.  geo: [(64 x 64 weight tiles, 16 x 8-pixel tiles)] per layer.
    A block of layer i walks ceil(t_i / s_i) pixel tiles (~2.3 us each) after ~9 us of fixed cost; with s_i > 1 it writes a 147 KB partial
    that the reduction reads back (s_i == 1: the kernel writes the gradient itself).  Blocks are dispatched longest first to the next free
    CU, one block per CU (147 KB of LDS).  Candidates: a common number of pixel tiles per block for grids of 192 ... 1 536 blocks; the
    cheapest simulated schedule wins."""
    import heapq
    work = float(sum(b * t for b, t in geo))
    best, best_cost = None, None
    forced = _WGRAD_DEFER_BLOCKS
    for target in ((forced,) if forced > 0 else range(192, 1537, 64)):
        per_block = max(4.0, work / target)
        ss = [max(1, min(int(t / per_block + 0.5), max(1, t // 4))) for b, t in geo]
        blocks = sorted(((9.0 + 2.3 * ((t + ns - 1) // ns), b * ns) for (b, t), ns in zip(geo, ss)), reverse=True)
        free = [0.0] * cus
        end = 0.0
        for d, cnt in blocks:
            for _ in range(cnt):
                t0 = heapq.heappop(free)
                heapq.heappush(free, t0 + d)
                end = max(end, t0 + d)
        cost = end + 0.0735 * sum(b * ns for (b, t), ns in zip(geo, ss) if ns > 1)      # one split: written as the gradient, no reduction
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = ss, cost
    return best


_ENCODERS = {
    "resnet18": dict(layers=(2, 2, 2, 2), out_channels=(3, 64, 64, 128, 256, 512)),
    "resnet34": dict(layers=(3, 4, 6, 3), out_channels=(3, 64, 64, 128, 256, 512)),
    "resnet50": dict(layers=(3, 4, 6, 3), out_channels=(3, 64, 256, 512, 1024, 2048), block="bottleneck"),
}


# ----------------------------------------------------------------------------------------------------------------
# execution
# ----------------------------------------------------------------------------------------------------------------
class _Unit:
    """One conv (+BatchNorm) node of the schedule."""

    def __init__(self, name, conv, bn, *, relu=True):
        self.name, self.conv, self.bn, self.relu = name, conv, bn, relu
        self.k = conv.kernel_size[0]
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.cin_p = (self.cin + 7) // 8 * 8
        self.cout_p = (self.cout + 7) // 8 * 8


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, hook, runner):
        out = runner.run_forward_train(x)
        ctx.runner = runner
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient w.r.t. the Unet input is not on the hot path (IR images are data)")
        ctx.runner.run_backward(dout)
        return None, None, None


class _RawAct:
    """The output of a Conv2dReLU unit BEFORE its BatchNorm + ReLU, together with the BatchNorm's (scale, shift): what a consumer
    with consumer-side BatchNorm (ops.conv2d / ops.wgrad `in_scale`) reads instead of the normalised activation, which is then
    never written to HBM (src/segmentation_models/base/modules.py:10-47 fused into the NEXT convolution's operand staging)."""
    __slots__ = ("y", "scale", "shift", "relu")

    def __init__(self, y, scale, shift, relu):
        self.y, self.scale, self.shift, self.relu = y, scale, shift, relu

    @property
    def shape(self):
        return self.y.shape

    def materialize(self):
        """The normalised activation as a tensor (hd_bn_apply): tests / diagnostics only, the training step never calls it."""
        return ops.bn_apply(self.y, self.scale, self.shift, relu=self.relu)


def _operand(t):
    """-> (tensor, in_scale, in_shift, in_relu) of a conv / wgrad input that may be a _RawAct."""
    if isinstance(t, _RawAct):
        return t.y, t.scale, t.shift, t.relu
    return t, None, None, True


def _small_conv_ok(cin, cout, dual):
    """Shapes hd_conv2d routes to the small-channel 3x3 kernel AND hd_wgrad to its small-channel kernel (conv3x3_small.hip /
    wgrad3x3_small.hip): the only ones that implement consumer-side BatchNorm."""
    return (not dual) and cin in (16, 32) and cout in (16, 32)


class UnetRunner:
    """Explicit forward/backward schedule over the C ABI for one Unet module."""

    def __init__(self, module):
        self.module = module
        self.grad_scale = 1.0          # loss scale S applied upstream; parameter gradients are emitted as g/S
        self.act_dtype = torch.float16 # storage type of activations / packed weights / gradient maps: set_precision(32) -> float32
        self.saved = None
        self.use_graphs = False        # replay the (static-shape) schedule as two hipGraphs: see enable_graphs()
        self._g = None
        # data parallelism: called as hook(lo, hi) when the gradient-arena slice [lo, hi) is final, decoder + head first, then
        # layer4 .. layer1 + stem (the arena is in parameter order, so backward completes it from the end).  With a hook the
        # backward graph is captured in SEGMENTS and the hook runs between their replays (the collective of bucket k then
        # overlaps the kernels of segment k+1).
        self.bucket_hook = None
        self._cut = None
        self._red = None
        self._wg_plan = None
        self._wg_ready = []
        self._static_douts = []
        self._wg_side = None           # side stream of the deferred weight gradients (_fork_wgrads)
        self._wg_forked = []           # operands of forked launches: kept alive until _join_wgrads (no reuse of their memory before)
        enc, dec = module.encoder, module.decoder
        self.stem = _Unit("encoder.conv1", enc.conv1, enc.bn1)
        self.stages = []
        for li in range(1, 5):
            blocks = []
            for bi, blk in enumerate(getattr(enc, "layer%d" % li)):
                pre = "encoder.layer%d.%d." % (li, bi)
                us = [_Unit(pre + "conv1", blk.conv1, blk.bn1), _Unit(pre + "conv2", blk.conv2, blk.bn2)]
                if hasattr(blk, "conv3"):             # Bottleneck (resnet50)
                    us.append(_Unit(pre + "conv3", blk.conv3, blk.bn3))
                ud = _Unit(pre + "downsample", blk.downsample[0], blk.downsample[1], relu=False) if blk.downsample is not None else None
                blocks.append((us, ud))
            self.stages.append(blocks)
        self.dec = []
        for i, b in enumerate(dec.blocks):
            pre = "decoder.blocks.%d." % i
            self.dec.append((_Unit(pre + "conv1", b.conv1[0], b.conv1[1]), _Unit(pre + "conv2", b.conv2[0], b.conv2[1]), b.in_channels, b.skip_channels))
        self.head_conv = module.segmentation_head[0]
        self.units = [self.stem] + [u for st in self.stages for (us, ud) in st for u in us + ([ud] if ud is not None else [])] + [u for d in self.dec for u in d[:2]]
        self._wplan = {}
        # Consumer-side BatchNorm: a decoder unit whose ONLY consumer is a small-channel 3x3 convolution (next conv of the decoder or
        # the segmentation head) hands on its raw conv output + BatchNorm coefficients; hd_bn_apply is not launched for it and the
        # four largest activations of the network (32 ch @ 256x320 x2, 16 ch @ 512x640 x2) are never written / re-read in normalised
        # form.  HD_BN_FUSE=0: every unit materialises its activation (A/B knob; the two forms are bit-identical, tested).
        self.fuse_bn = os.environ.get("HD_BN_FUSE", "1") != "0"
        # BatchNorm backward sums from the epilogue of the data-gradient kernel that produces the unit's incoming gradient
        # (hd_conv_args.bs_*: 8-wave 3x3 kernels); HD_BN_BWD_SUMS=0: always the separate reduction pass (A/B)
        self.fuse_bwd_sums = os.environ.get("HD_BN_BWD_SUMS", "1") != "0"
        self._raw_units = set()
        nd = len(self.dec)
        for i, (u1, u2, cin, cskip) in enumerate(self.dec):
            if _small_conv_ok(u1.cout, u2.cout, False):
                self._raw_units.add(u1.name)                        # consumer: conv2 of the same block
            if i + 1 < nd:
                n1 = self.dec[i + 1]
                if _small_conv_ok(u2.cout, n1[0].cout, n1[3] > 0):
                    self._raw_units.add(u2.name)                    # consumer: conv1 of the next block (no skip source)
            elif u2.cout == 16 and self.head_conv.out_channels <= 16:
                self._raw_units.add(u2.name)                        # consumer: the segmentation head (16 -> <= 16 channels)

    def saved_activations(self):
        """{unit name: post-BatchNorm(+ReLU) activation, NHWC f16} of the last saved training forward (tests: ReLU decisions for the
        oracle).  Units that hand on their raw output (consumer-side BatchNorm) are normalised here, on demand."""
        return {k: (v["z"].materialize() if isinstance(v["z"], _RawAct) else v["z"]) for k, v in self.saved["rec"].items()}

    # ------------------------------------------------------------------ hipGraph replay
    def enable_graphs(self, on=True):
        """The training forward and backward are fixed launch sequences over fixed buffers (~650 + ~700 launches at
        3-30 us of GPU time each): capture each once per input shape / loss scale and replay it, which removes the
        per-launch host cost (the step is host-bound otherwise).  Eager execution stays the default."""
        self.use_graphs = on
        self._g = None

    @staticmethod
    def _copy_input(static, x):
        """x -> the forward graph's static input; a stride-0 channel view (1 -> 3 channel `expand`) holds ONE plane per image."""
        if static.stride(1) == 0:
            static[:, :1].copy_(x[:, :1])
        else:
            static.copy_(x)

    def run_forward_train(self, x):
        if not self.use_graphs:
            return self.forward(x, training=True, save=True)
        g = self._g
        if g is None or g["xshape"] != tuple(x.shape) or g["dev"] != x.device or g.get("xkind") != (x.stride(1) == 0):
            self.flatten_parameters()
            if x.stride(1) == 0 and ops.dense_planes(x):      # stride-0 channel view (IR image): the static input is ONE plane per image
                x_static = torch.empty((x.shape[0], 1) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device).expand(-1, x.shape[1], -1, -1)
            else:
                x_static = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
            g = self._g = dict(xshape=tuple(x.shape), dev=x.device, x=x_static, bwd=None, scale=None, xkind=x.stride(1) == 0)
            self._copy_input(g["x"], x)
            g["xsrc"] = None
            snap = [b.clone() for b in self.module.buffers()]   # the warm-up run must not count as a training step
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up outside capture (lazy module loading, allocator)
                self.forward(g["x"], training=True, save=True)
            torch.cuda.current_stream().wait_stream(side)
            g["pool"] = torch.cuda.graph_pool_handle()
            g["fwd"] = torch.cuda.CUDAGraph()
            # thread_local: the input pipeline's helper thread (DevicePrefetcher) allocates pinned / device memory while this thread
            # captures; in the default 'global' mode such a call from ANOTHER thread invalidates the capture
            with capture_without_gc(), torch.cuda.graph(g["fwd"], pool=g["pool"], capture_error_mode="thread_local"):
                g["out"] = self.forward(g["x"], training=True, save=True)
            g["saved"] = self.saved
            # the capture itself did not execute: restore the BatchNorm buffers the warm-up touched, then replay
            for b, s0 in zip(self.module.buffers(), snap):
                b.copy_(s0)
        # a batch that is the very tensor copied last time, unmodified since (a resident batch re-used step after step), is not copied
        # again -- the rule the detector graph's staging follows; the reference held in g["xsrc"] keeps its storage from being recycled
        base = x._base if x._base is not None else x
        src = g.get("xsrc")
        if not (src is not None and src[0] is base and src[1] == base._version and src[2] == x.data_ptr() and src[3] == x.stride()):
            self._copy_input(g["x"], x)
            g["xsrc"] = (base, base._version, x.data_ptr(), x.stride())
        g["fwd"].replay()
        self.saved = g["saved"]
        return g["out"]

    def bucket_ranges(self):
        """[(lo, hi)] slices of the flat gradient arena in the order the backward pass completes them.  Cached per arena: the hooked
        backward asks every step, and walking ~180 parameters in Python between the detector graph and the first backward graph was a
        160 us hole in every data-parallel step (profiles/r06_forced_dist_steady_state.txt, first collection)."""
        self.flatten_parameters()
        key = (self._gflat.data_ptr(), self._gflat.numel(), _EXCHANGE_BUCKETS)
        if getattr(self, "_bucket_ranges_cache", (None, None))[0] == key:
            return self._bucket_ranges_cache[1]
        out = self._bucket_ranges_uncached()
        self._bucket_ranges_cache = (key, out)
        return out

    def _bucket_ranges_uncached(self):
        off = {id(p): o for p, o in zip(self._params, self._offsets)}
        total = self._gflat.numel()
        first = lambda mod: min(off[id(p)] for p in mod.parameters())
        enc = self.module.encoder
        if _EXCHANGE_BUCKETS == 2:
            # two buckets (round 5): decoder + layer4 + layer3 (94 % of a resnet34 U-Net's parameters, final after 60 % of the pass) and
            # the rest -- the boundaries at which the deferred weight gradients are launched as whole groups (_segment_done)
            cuts = [first(enc.layer3), 0]
        else:
            cuts = [first(self.module.decoder)] + [first(getattr(enc, "layer%d" % li)) for li in (4, 3, 2)] + [0]
        out, hi = [], total
        for lo in cuts:
            out.append((lo, hi))
            hi = lo
        return out

    def _reduce(self, slab, dw, KH, KW, Cin, **kw):
        """hd_wgrad_reduce of one conv's slab: deferred to the end of the backward segment, where all of the segment's weight
        tensors are summed in ONE launch (ops.WgradReduceBatch; HD_WRED_MULTI=0: one launch per conv, as before)."""
        if self._red is None:
            ops.wgrad_reduce(slab, dw, KH, KW, Cin, **kw)
        else:
            self._red.add(slab, dw, KH, KW, Cin, **kw)

    def _segment_done(self, k):
        """Called by backward() at segment boundary k (0 = decoder + head done, 1..3 = layer4..layer2, 4 = everything).
        Deferred 8-wave weight gradients (see _unit_bwd) are launched as two groups, after segment 2 (decoder + layer4 + layer3) and
        after segment 4 (layer2 + layer1): a grid of three stages' layers keeps the chip full where one stage's 192 or 384 equal blocks
        leave a quarter of it idle.  A gradient-exchange hook sees the same two boundaries (HD_EXCHANGE_BUCKETS=2: bucket_ranges() has two
        entries, the backward pass is two hipGraphs); HD_EXCHANGE_BUCKETS=5: one bucket, one launch group and one graph per segment."""
        hooked = self._cut is not None or self.bucket_hook is not None
        per_segment = (hooked and _EXCHANGE_BUCKETS != 2) or not _WGRAD_MERGE
        act = per_segment or k in (2, 4)
        if self._wg_ready and act:
            if _WGRAD_SIDE in (1, 2) and not hooked and not per_segment and k == 2:
                self._fork_wgrads()
            else:
                self._launch_wgrads()
        if k == 4:
            self._join_wgrads()
        if self._red is not None and (act or not hooked):
            self._red.flush()                   # the segment's gradients are final only after this launch
        if not hooked or (_EXCHANGE_BUCKETS == 2 and k not in (2, 4)):
            return
        if self._cut is not None:
            self._cut(k)                        # (graph capture: a new graph begins behind every boundary but the last)
        else:
            self.bucket_hook(*self.bucket_ranges()[(0 if k == 2 else 1) if _EXCHANGE_BUCKETS == 2 else k])

    def adopt_static_dout(self, t):
        """`t` is a buffer some producer rewrites in place before every backward pass (the detector graph's image gradient): if the
        gradient that arrives when the backward graphs are captured IS this buffer, they read it in place.  The list keeps it alive."""
        self._static_douts = [u for u in self._static_douts if u.data_ptr() != t.data_ptr()][-7:] + [t]

    def run_backward(self, dout):
        if not self.use_graphs:
            return self.backward(dout.contiguous())
        g = self._g
        S = float(self.grad_scale)
        segmented = self.bucket_hook is not None
        if g["bwd"] is None or g["scale"] != S or g.get("segmented") != segmented:
            adopted = [t for t in self._static_douts if t.data_ptr() == dout.data_ptr() and t.shape == g["out"].shape and t.dtype == torch.float32 and t.is_contiguous()]
            if adopted:
                g["dout"] = adopted[0]          # a producer's static output (det_graph.py): read in place, no copy per step
            else:
                g["dout"] = torch.empty(g["out"].shape, dtype=torch.float32, device=g["dev"])
                g["dout"].copy_(dout)
            self.saved = g["saved"]
            g["bwd"] = None                 # a failed capture must not leave graphs of another (scale, segmentation) behind
            torch.cuda.synchronize()
            if not segmented:
                graphs = [torch.cuda.CUDAGraph()]
                with capture_without_gc(), torch.cuda.graph(graphs[0], pool=g["pool"], capture_error_mode="thread_local"):
                    self.backward(g["dout"], keep_saved=True)
            else:
                # one graph per bucket: end the running capture at every boundary and begin the next one in the same pool
                graphs = [torch.cuda.CUDAGraph()]
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                hook, self.bucket_hook = self.bucket_hook, None
                with capture_without_gc(), torch.cuda.stream(side):
                    graphs[0].capture_begin(pool=g["pool"], capture_error_mode="thread_local")

                    def cut(k):
                        if k < 4:
                            graphs[-1].capture_end()
                            graphs.append(torch.cuda.CUDAGraph())
                            graphs[-1].capture_begin(pool=g["pool"], capture_error_mode="thread_local")
                    self._cut = cut
                    err = None
                    try:
                        self.backward(g["dout"], keep_saved=True)
                    except BaseException as e:          # keep the ORIGINAL error: ending an invalidated capture raises its own
                        err = e
                    finally:
                        self._cut = None
                        self.bucket_hook = hook
                        try:
                            graphs[-1].capture_end()
                        except Exception:
                            if err is None:
                                raise
                    if err is not None:
                        raise err
                torch.cuda.current_stream().wait_stream(side)
            g["bwd"], g["scale"], g["segmented"] = graphs, S, segmented       # only a complete capture is remembered
        if dout.data_ptr() != g["dout"].data_ptr():     # (det_graph.py writes the image gradient straight into the static input)
            g["dout"].copy_(dout)
        if not segmented:
            g["bwd"][0].replay()
        else:
            ranges = self.bucket_ranges()
            for k, gr in enumerate(g["bwd"]):
                gr.replay()
                self.bucket_hook(*ranges[k])
        return None

    # ------------------------------------------------------------------ parameters
    def flatten_parameters(self):
        """Make every parameter (and gradient) a view of one fp32 arena; BN running stats likewise."""
        if getattr(self, "_flat", None) is not None:
            # cheap aliasing probe (first / last parameter and gradient still views of the arenas)
            ps, o = self._params, self._offsets
            if all(p.data_ptr() == self._flat.data_ptr() + k * 4 and p.grad is not None and p.grad.data_ptr() == self._gflat.data_ptr() + k * 4
                   for p, k in ((ps[0], o[0]), (ps[-1], o[-1]))):
                return
        params = [p for p in self.module.parameters()]
        dev = params[0].device
        if getattr(self, "_flat", None) is not None and self._flat.device == dev and all(
                p.data_ptr() == self._flat.data_ptr() + o * 4 and p.grad is not None
                and p.grad.data_ptr() == self._gflat.data_ptr() + o * 4 for p, o in zip(params, self._offsets)):
            return
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4     # keep every view 16-byte aligned
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = gflat[o:o + p.numel()].view(p.shape)
        self._flat, self._gflat, self._offsets = flat, gflat, offs
        self._params = params

    @property
    def flat_params(self):
        self.flatten_parameters()
        return self._flat

    @property
    def flat_grads(self):
        self.flatten_parameters()
        return self._gflat

    def set_precision(self, precision):
        """16: fp16 storage (BASELINE configs[1]); 32: fp32 storage, the arithmetic of the reference's default `--precision 32`
        (src/config/config.py:149) -- every kernel of the schedule in its _f32 form (parity mode: not a fast path)."""
        dt = torch.float32 if int(precision) == 32 else torch.float16
        if dt != self.act_dtype:
            self.act_dtype = dt
            self._wplan = {}
            self._g = None
            self.saved = None

    def _prep_weights(self, need_dgrad):
        """fp32 OIHW masters -> fp16 GEMM layouts (every step in training: the masters move)."""
        hc = self.head_conv
        params = [u.conv.weight for u in self.units] + [hc.weight]
        plan = self._wplan.get(need_dgrad)
        if plan is None or not plan.valid_for(params):
            # one launch for every layer (the masters are views of the flat arena: stable pointers -> the table is built once)
            items = [(u.conv.weight, u.cin_p, u.cout_p, need_dgrad and u is not self.stem) for u in self.units]
            items.append((hc.weight, hc.in_channels, 8, need_dgrad))
            plan = self._wplan[need_dgrad] = ops.WeightPrepPlan(items, dtype=self.act_dtype)
        outs = plan.run()
        W = {u.name: outs[i] for i, u in enumerate(self.units)}
        W["head"] = outs[-1]
        return W

    # ------------------------------------------------------------------ forward pieces
    def _conv_bn(self, u, x, W, training, rec, *, x2=None, up1=False, res=None):
        """conv -> BatchNorm(train: batch statistics / eval: running statistics) (+res) (+ReLU).  Returns z -- or, for the units
        in `_raw_units`, a _RawAct (raw conv output + BatchNorm coefficients) that the consumer normalises on the fly.  `x` may
        itself be a _RawAct."""
        wf = W[u.name][0]
        xin = x
        x, isc, ish, irelu = _operand(xin)
        bnk = dict(in_scale=isc, in_shift=ish, in_relu=irelu) if isc is not None else {}
        if training:
            y, stats = ops.conv2d(x, wf, u.k, u.k, x2=x2, stride=u.stride, pad=u.pad, up1=up1, want_stats=True, **bnk)
            npix = y.numel() // u.cout
            mean, invstd, scale, shift = ops.bn_finalize(stats, npix, u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var,
                                                         u.bn.momentum, u.bn.eps)
        else:
            y = ops.conv2d(x, wf, u.k, u.k, x2=x2, stride=u.stride, pad=u.pad, up1=up1, **bnk)
            scale, shift = ops.bn_eval_scale_shift(u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var, u.bn.eps)
            mean = invstd = None
        if self.fuse_bn and res is None and u.name in self._raw_units:
            z = _RawAct(y, scale, shift, u.relu)
        else:
            z = ops.bn_apply(y, scale, shift, res=res, relu=u.relu)
        if rec is not None:
            rec[u.name] = dict(x=xin, x2=x2, up1=up1, y=y, z=z, mean=mean, invstd=invstd, has_res=res is not None)
        return z

    def forward(self, x, training, save):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("Unet expects [N,3,H,W], got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise RuntimeError("hallucidet_amd Unet runs on the GPU only (input is on %s); there is no CPU path" % x.device)
        self.flatten_parameters()
        N, _, H, Wd = x.shape
        save = save and training
        rec = {} if save else None
        Wt = self._prep_weights(need_dgrad=save)
        x = ops.as_dense_planes_f32(x)              # a 1 -> 3 channel `expand` view stays a view (read three times by the kernel below)
        a0 = ops.nchw_to_nhwc_resize(x, H, Wd, 8, dtype=self.act_dtype)
        f1 = self._conv_bn(self.stem, a0, Wt, training, rec)
        cur, pool_idx = ops.maxpool3x3s2_idx(f1) if save else (ops.maxpool3x3s2(f1), None)
        feats = [f1]
        pooled = cur
        for blocks in self.stages:
            for (us, ud) in blocks:
                z = cur
                for u in us[:-1]:
                    z = self._conv_bn(u, z, Wt, training, rec)
                idt = cur if ud is None else self._conv_bn(ud, cur, Wt, training, rec)
                cur = self._conv_bn(us[-1], z, Wt, training, rec, res=idt)
            feats.append(cur)
        # feats = [f1, f2, f3, f4, f5]
        skips = [feats[3], feats[2], feats[1], feats[0], None]
        d = feats[4]
        for (u1, u2, cin, cskip), skip in zip(self.dec, skips):
            z1 = self._conv_bn(u1, d, Wt, training, rec, x2=skip, up1=True)
            d = self._conv_bn(u2, z1, Wt, training, rec)
        hc = self.head_conv
        dt, isc, ish, irelu = _operand(d)
        out = ops.conv2d(dt, Wt["head"][0], 3, 3, bias=hc.bias, pad=1, act=ACT_SIGMOID, out_nchw_f32=True, cout=hc.out_channels,
                         **(dict(in_scale=isc, in_shift=ish, in_relu=irelu) if isc is not None else {}))
        if training:
            self._bump_batches_tracked()
        if save:
            self.saved = dict(rec=rec, W=Wt, f1=f1, pool_idx=pool_idx, pooled=pooled, feats=feats, head_in=d, out=out, shape=(N, H, Wd))
        return out

    def _bump_batches_tracked(self):
        """BatchNorm2d.num_batches_tracked += 1 for every unit: the counters are views of ONE int64 buffer (same keys and
        values in state_dict), so a step bumps them with one launch instead of one per BatchNorm (47 launches of ~4 us)."""
        bns = [u.bn for u in self.units if u.bn.num_batches_tracked is not None]
        if not bns:
            return
        flat = getattr(self, "_nbt_flat", None)
        dev = bns[0].num_batches_tracked.device
        if flat is None or flat.device != dev or flat.numel() != len(bns) or any(
                b.num_batches_tracked.data_ptr() != flat.data_ptr() + 8 * i for i, b in enumerate(bns)):
            flat = torch.stack([b.num_batches_tracked.detach().reshape(()).to(torch.int64) for b in bns])
            for i, b in enumerate(bns):
                b._buffers["num_batches_tracked"] = flat[i]
            self._nbt_flat = flat
        flat += 1

    # ------------------------------------------------------------------ backward pieces
    def _bstat_of(self, v):
        """conv2d(bstat=...) request for unit `v`: the data gradient about to be computed is v's incoming gradient, so the kernel that
        writes it can emit v's BatchNorm backward sums (hd_conv_args.bs_*); None when v's output did not reach the consumer as a plain
        materialised activation (consumer-side BatchNorm units keep their own reduction)."""
        if v is None or not self.fuse_bwd_sums:
            return None
        r = self.saved["rec"][v.name]
        if isinstance(r["z"], _RawAct):
            return None
        return dict(y=r["y"], z=r["z"] if r["has_res"] else None, mean=r["mean"], invstd=r["invstd"], gamma=v.bn.weight, beta=v.bn.bias, relu=v.relu)

    def _unit_bwd(self, u, dz, S, *, want_dres=False, need_dx=True, dx_res=None, part=None, dx_is_dz_of=None, pool2=None):
        """Backward through z = relu(bn(conv(x)) (+res)).  Writes dW/dgamma/dbeta into the flat gradient arena.
        `part`: the BatchNorm reduction rows of THIS unit when the kernel that produced dz emitted them; `dx_is_dz_of`: the unit whose
        output is this convolution's input and receives no other gradient -- its reduction rows are requested from the data-gradient
        kernel.  Returns (dx over the logical conv input or None, dres or None, rows for dx_is_dz_of or None)."""
        r = self.saved["rec"][u.name]
        inv = 1.0 / S
        # the saved activation is only needed as a ReLU mask when a residual was added; otherwise it is recomputed from y
        dy, dres, _, _ = ops.bn_backward(dz, r["z"] if r["has_res"] else None, r["y"], r["mean"], r["invstd"], u.bn.weight, u.bn.bias,
                                         relu=u.relu, want_dres=want_dres,
                                         gscale=inv, dgamma=u.bn.weight.grad, dbeta=u.bn.bias.grad, part=part)
        xt, isc, ish, irelu = _operand(r["x"])
        wkw = dict(x2=r["x2"], stride=u.stride, pad=u.pad, up1=r["up1"], **(dict(in_scale=isc, in_shift=ish, in_relu=irelu) if isc is not None else {}))
        dx = None
        bstat = None
        if need_dx:
            # weight gradient and data gradient read the same dY and are independent: one call, ONE grid where both run in the 8-wave
            # kernels (ops.wgrad_dgrad / hd_conv2d_wgrad: the deep layers' 160-tile data gradients leave 96 CUs idle on their own)
            wd = self.saved["W"][u.name][1]
            x = xt
            if r["up1"]:
                hw = (x.shape[1] * 2, x.shape[2] * 2)
            else:
                hw = (x.shape[1], x.shape[2])
            bstat = self._bstat_of(dx_is_dz_of) if (u.stride == 1 and not r["up1"] and r["x2"] is None) else None
            dkw = dict(stride=1, pad=u.k - 1 - u.pad, in_dil=u.stride, out_hw=hw, cout=u.cin_p, res=dx_res,
                       **(dict(bstat=bstat) if bstat is not None else {}), **(dict(pool2=pool2) if pool2 is not None else {}))
            if self._wg_plan is not None and u.name in self._wg_plan:
                # the weight gradient waits for the end of the backward segment, where all of the segment's 8-wave weight gradients are ONE
                # grid (ops.wgrad_multi): a block per 64 x 64 weight tile and LAYER walks that layer's pixel tiles and writes one 147 KB
                # partial, where 256 blocks per layer wrote 256 -- 1 / 16 ... 1 / 256 of the slab bytes, the per-block fixed cost once.
                dx = ops.conv2d(dy, wd, u.k, u.k, **dkw)
                ns = self._wg_plan[u.name]
                tiles = dy.shape[0] * ((dy.shape[1] + 15) // 16) * ((dy.shape[2] + 7) // 8)
                self._wg_ready.append(((u, xt, dy, wkw, inv), ns, (tiles + ns - 1) // ns))
                return dx, dres, (bstat["part"] if bstat is not None else None)
            slab, dx = ops.wgrad_dgrad(xt, dy, u.k, u.k, wd, dgrad=dkw, **wkw)
        else:
            slab = ops.wgrad(xt, dy, u.k, u.k, **wkw)
        self._reduce(slab, u.conv.weight.grad, u.k, u.k, u.cin_p, Cin_real=u.cin, scale=inv)
        return dx, dres, (bstat["part"] if bstat is not None else None)

    def _plan_wgrads(self):
        """{unit name: pixel split} for every convolution whose weight gradient the backward pass defers (3x3 / stride 1, >= 64 channels on
        both sides, materialised operand: the 8-wave kernel's domain), from the saved forward record -- BEFORE the pass, for the two launch
        groups (segments 0-2: decoder, layer4, layer3; segments 3-4: layer2, layer1) as whole grids, so that a layer's split, and with it
        the rounding of its gradient, is the same whether the groups are launched whole (no exchange hook) or segment by segment."""
        rec = self.saved["rec"]
        seg_units = [[u for d in self.dec for u in d[:2]]] + [[u for (us, ud) in self.stages[si] for u in us + ([ud] if ud is not None else [])] for si in (3, 2, 1, 0)]
        # The plan depends on shapes only (operand extents per unit + which operands are materialised): ~40 ctypes probes and a heap
        # simulation of 22 candidate grids per group cost ~14 ms of host time (more than a whole graphed step), so it is computed once
        # per geometry -- the eager runner (use_graphs=False, every test that calls net(x).backward) and every re-capture hit the cache.
        cand = []
        for k in range(5):
            for u in seg_units[k]:
                r = rec[u.name]
                xt, isc, _, _ = _operand(r["x"])
                if isc is not None or u.k != 3 or u.stride != 1:
                    continue
                cand.append((k, u, r, xt))
        key = tuple((u.name, tuple(xt.shape), tuple(r["y"].shape), None if r["x2"] is None else tuple(r["x2"].shape), bool(r["up1"]), xt.dtype)
                    for _, u, r, xt in cand) + (_WGRAD_DEFER_BLOCKS,)
        cache = self.__dict__.setdefault("_wg_plan_cache", {})
        if key in cache:
            return cache[key]
        plan = {}
        for group in ((0, 1, 2), (3, 4)):
            names, geo = [], []
            for k, u, r, xt in cand:
                if k not in group:
                    continue
                y = r["y"]
                b = ops.wgrad_w8_blocks(xt, y, u.k, u.k, x2=r["x2"], stride=u.stride, pad=u.pad, up1=r["up1"])
                if b > 0:
                    names.append(u.name)
                    geo.append((b, y.shape[0] * ((y.shape[1] + 15) // 16) * ((y.shape[2] + 7) // 8)))
            if names:
                plan.update(zip(names, _plan_wgrad_splits(geo)))
        if len(cache) >= 8:
            cache.pop(next(iter(cache)))
        cache[key] = plan
        return plan

    def _fork_wgrads(self):
        """The ready weight gradients (and the reduction of their slabs) on the side stream, behind everything the current stream has
        been given so far; the current stream goes on with the next segment's chain.  Their operands stay referenced until the join, so
        the allocator (eager or a graph's private pool) cannot hand their memory to a later tensor of the main chain."""
        if not self._wg_ready:
            return
        main = torch.cuda.current_stream()
        if self._wg_side is None:
            self._wg_side = torch.cuda.Stream()
        side = self._wg_side
        self._wg_forked.append(list(self._wg_ready))
        side.wait_stream(main)
        red, self._red = self._red, (ops.WgradReduceBatch() if _WRED_MULTI else None)
        with torch.cuda.stream(side):
            self._launch_wgrads()
            if self._red is not None:
                self._red.flush()
        self._red = red

    def _join_wgrads(self):
        if self._wg_forked:
            torch.cuda.current_stream().wait_stream(self._wg_side)
            self._wg_forked = []

    def _launch_wgrads(self):
        """The planned weight gradients as one grid per <= 24 layers, longest blocks first.  One pixel split: the kernel writes the scaled
        OIHW gradient itself (hd_wgrad_args.dw_oihw), no slab, no reduction."""
        ready, self._wg_ready = self._wg_ready, []
        ready.sort(key=lambda e: -e[2])
        for lo in range(0, len(ready), ops.WGRAD_MULTI_MAX):
            part = ready[lo:lo + ops.WGRAD_MULTI_MAX]
            calls = [(xt, dy, u.k, u.k, dict(nsplit=ns, **(dict(dw=u.conv.weight.grad, dw_scale=inv) if (ns == 1 and _WGRAD_DIRECT and u.cin_p == u.cin) else {}), **wkw))
                     for (u, xt, dy, wkw, inv), ns, _ in part]
            for ((u, xt, dy, wkw, inv), ns, _), slab in zip(part, ops.wgrad_multi(calls)):
                if slab is not None:
                    self._reduce(slab, u.conv.weight.grad, u.k, u.k, u.cin_p, Cin_real=u.cin, scale=inv)

    def backward(self, dout, need_dx=False, keep_saved=False):
        sv = self.saved
        if sv is None:
            raise RuntimeError("Unet.backward called without a saved training forward")
        S = float(self.grad_scale)
        inv = 1.0 / S
        N, H, Wd = sv["shape"]
        hc = self.head_conv
        self._red = ops.WgradReduceBatch() if _WRED_MULTI else None
        self._wg_plan = self._plan_wgrads() if (_WGRAD_DEFER and self.act_dtype == torch.float16) else None
        self._wg_ready = []
        # head: sigmoid' then conv backward
        dl = ops.sigmoid_bwd_nchw_to_nhwc(dout.float(), sv["out"], 8, 1.0, dtype=self.act_dtype)
        db = ops.channel_sum(dl)
        ops.scale_store(db, hc.bias.grad, inv, accumulate=False)
        ht, isc, ish, irelu = _operand(sv["head_in"])
        slab = ops.wgrad(ht, dl, 3, 3, pad=1, **(dict(in_scale=isc, in_shift=ish, in_relu=irelu) if isc is not None else {}))
        self._reduce(slab, hc.weight.grad, 3, 3, hc.in_channels, Cout=hc.out_channels, scale=inv)
        dz = ops.conv2d(dl, sv["W"]["head"][1], 3, 3, pad=1, cout=hc.in_channels)
        # decoder, last block first
        feats = sv["feats"]
        skips = [feats[3], feats[2], feats[1], feats[0], None]
        dskip = [None] * 5
        for i in range(len(self.dec) - 1, -1, -1):
            u1, u2, cin, cskip = self.dec[i]
            dz1, _, rows1 = self._unit_bwd(u2, dz, S, dx_is_dz_of=u1)
            # the gradient of the block's low-resolution input is the 2x2 sum of the upsampled half's data gradient -- asked from the
            # data-gradient kernel's epilogue (ops.conv2d pool2); the full-resolution concatenated gradient is then never written
            # (with a skip: the 32 -> 128-channel kernel of block 3 writes the pooled half and the skip's half itself; elsewhere the
            #  request is declined and hd_concat_up_bwd runs as before)
            pool = dict(c_up=cin) if _POOL2 else None
            dcat, _, _ = self._unit_bwd(u1, dz1, S, part=rows1, pool2=pool)
            if pool is not None and pool.get("done"):
                dz, dskip[i] = dcat, pool.get("skip")
                continue
            dz, dskip[i] = ops.concat_up_bwd(dcat, cin)        # 2x2 sum-pool of the upsampled half + the skip's slice, one launch
        self._segment_done(0)                  # head + decoder parameter gradients are final
        # dz is now the gradient of f5; dskip[0..3] belong to f4, f3, f2, f1
        dfeat = {4: dz, 3: dskip[0], 2: dskip[1], 1: dskip[2], 0: dskip[3]}
        d_out, rows_out = dfeat[4], None          # rows_out: BatchNorm reduction rows of the unit whose output gradient d_out is
        for si in range(3, -1, -1):
            blocks = self.stages[si]
            for bi in range(len(blocks) - 1, -1, -1):
                us, ud = blocks[bi]
                # gradient that the block INPUT also receives from elsewhere (decoder skip) when it is a stage output
                extra = dfeat[si] if bi == 0 and si > 0 else None
                # the unit below in the chain receives its whole gradient from this unit's data gradient
                dz1, dres, rows = self._unit_bwd(us[-1], d_out, S, want_dres=True, part=rows_out, dx_is_dz_of=us[-2] if len(us) > 1 else None)
                for j in range(len(us) - 2, 0, -1):          # Bottleneck's middle 3x3
                    dz1, _, rows = self._unit_bwd(us[j], dz1, S, part=rows, dx_is_dz_of=us[j - 1])
                # the block input is the previous block's output (same stage, bi > 0): with the identity path and the skip folded in
                # through dx_res, this data gradient is that unit's complete incoming gradient
                prev_out = blocks[bi - 1][0][-1] if bi > 0 else None
                if ud is None:
                    acc = dres if extra is None else ops.add_f16(dres, extra)
                    d_in, _, rows_out = self._unit_bwd(us[0], dz1, S, dx_res=acc, part=rows, dx_is_dz_of=prev_out)
                else:
                    d_in, _, _ = self._unit_bwd(us[0], dz1, S, dx_res=extra, part=rows)
                    d_in, _, rows_out = self._unit_bwd(ud, dres, S, dx_res=d_in, dx_is_dz_of=prev_out)
                d_out = d_in
            if si > 0:
                self._segment_done(4 - si)     # layer4 -> 1, layer3 -> 2, layer2 -> 3
        # d_out = gradient of the max-pooled stem output
        if _WGRAD_SIDE in (1, 3) and self._cut is None and self.bucket_hook is None and _WGRAD_MERGE:
            self._fork_wgrads()                # layer2 + layer1 weight gradients beside the max-pool / stem backward
        df1 = ops.maxpool3x3s2_bwd_idx(sv["pool_idx"], d_out, (sv["f1"].shape[1], sv["f1"].shape[2]), add=dfeat[0])
        dx = None
        if need_dx:
            raise NotImplementedError("gradient w.r.t. the Unet input is not on the hot path (IR images are data)")
        self._unit_bwd(self.stem, df1, S, need_dx=False)          # (rows_out is None here: layer1.0's input is the max-pooled stem)
        self._segment_done(4)                  # layer1 + stem
        if not keep_saved:
            self.saved = None
        return dx


class SegmentationModel(nn.Module):
    def check_input_shape(self, x):
        """base/model.py:12-22 (same message)."""
        h, w = x.shape[-2:]
        s = self.encoder.output_stride
        if h % s != 0 or w % s != 0:
            new_h = (h // s + 1) * s if h % s != 0 else h
            new_w = (w // s + 1) * s if w % s != 0 else w
            raise RuntimeError(
                f"Wrong input shape height={h}, width={w}. Expected image height and width "
                f"divisible by {s}. Consider pad your images to shape ({new_h}, {new_w})."
            )


class Unet(SegmentationModel):
    """`smp.Unet(encoder_name, encoder_depth, encoder_weights, ..., in_channels, classes)` for resnet18/34/50 encoders."""

    def __init__(self, encoder_name="resnet34", encoder_depth=5, encoder_weights=None, decoder_use_batchnorm=True,
                 decoder_channels=(256, 128, 64, 32, 16), decoder_attention_type=None, in_channels=3, classes=1,
                 activation=None, aux_params=None):
        super().__init__()
        if encoder_name not in _ENCODERS:
            raise KeyError("Wrong encoder name `{}`, supported encoders: {}".format(encoder_name, list(_ENCODERS)))
        if encoder_depth != 5 or in_channels != 3 or aux_params is not None:
            raise NotImplementedError("hallucidet_amd Unet: encoder_depth=5, in_channels=3, aux_params=None (reference hot path)")
        if isinstance(encoder_weights, str) and encoder_weights not in ("imagenet",):
            # a local state_dict path
            sd = torch.load(encoder_weights, map_location="cpu")
        else:
            sd = None   # 'imagenet' cannot be downloaded offline: random initialisation (SURVEY App. D.8 deviation)
        self.encoder = ResNetEncoder(**_ENCODERS[encoder_name])
        if sd is not None:
            sd.pop("fc.bias", None)
            sd.pop("fc.weight", None)
            self.encoder.load_state_dict(sd)
        self.decoder = UnetDecoder(self.encoder.out_channels, decoder_channels, n_blocks=encoder_depth,
                                   use_batchnorm=decoder_use_batchnorm, center=False, attention_type=decoder_attention_type)
        self.segmentation_head = SegmentationHead(decoder_channels[-1], classes, kernel_size=3, activation=activation)
        self.classification_head = None
        self.name = "u-{}".format(encoder_name)
        initialize_decoder(self.decoder)
        initialize_head(self.segmentation_head)
        self._runner = None
        self._hook = None

    @property
    def runner(self):
        if self._runner is None:
            self._runner = UnetRunner(self)
        return self._runner

    def forward(self, x):
        self.check_input_shape(x)
        if not isinstance(self.segmentation_head[-1], nn.Sigmoid):
            raise NotImplementedError("hallucidet_amd Unet: the fused head kernel implements the 'sigmoid' head "
                                      "(Config.EncoderDecoder.decoder_head, config.py:78)")
        r = self.runner
        if not (self.training and torch.is_grad_enabled()):
            return r.forward(x, training=self.training, save=False)
        if self._hook is None or self._hook.device != x.device:
            self._hook = torch.zeros(1, device=x.device, requires_grad=True)
        return _UnetFn.apply(x, self._hook, r)
