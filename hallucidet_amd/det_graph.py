"""The detector half of a HalluciDet training step as ONE hipGraph replay.

What is captured (train_hallucidet.py:180-210 of the reference, plus the backward pass of :448-451 down to the hallucinated image):
the transform, the ResNet-50-FPN trunk and the heads over the hallucinated / RGB / IR batches, target assignment, the samplers'
random draws (the default generator is graph-safe: every replay advances its Philox offset), the losses and their weighting, the
loss-scaled backward pass through the frozen detector to dL/d(hallucinated image), and the deferred post-processing of the three
passes' detections.  Issued eagerly these are ~700 launches per step of 2-30 us each; the host cannot keep that queue full (rocprofv3
kernel trace, tools/trace_gaps.py: 0.7-1.5 ms of GPU idle time per 12.7 ms step, box dependent), a graph replay can.

Static shapes: the images are copied into static buffers; the targets are STAGED -- every image gets G box rows (G = the batch's
largest count rounded up to a multiple of 8), real boxes first, all-zero rows after them, plus a `_rows` mask for the reference's
degenerate-box check.  All-zero rows are what `detection.pad_targets` produces for ragged lists anyway (valid = x2 > x1), so the
arithmetic downstream is that of the eager path on the same staged targets: `tests/test_det_graph_gpu.py` holds replay == eager
bit for bit.  One graph per (shapes, G, loss weights); they share a memory pool (only one runs at a time).

The U-Net stays outside: its output enters through `_GraphedDetector` (an autograd Function whose backward hands the captured
dL/dimage to the U-Net runner's own backward graphs), so `scaler.scale(loss).backward()` in fit_step works unchanged.
"""
import os
import warnings

import torch

from .utils import eval_forward_fasterrcnn as _eff
from .models.detection import LazyDetections, _PAD_IDX_CACHE
from .segmentation_models.unet import capture_without_gc


def _bucket(n):
    """Rows per image of the staged targets: the next power of two >= 8.  Coarse on purpose -- every new bucket costs two eager warm-up
    bodies plus a capture in the middle of an epoch (and, under data parallelism, makes the other ranks wait in the all-reduce); capture
    steps also advance the samplers' generator further than a replay does, so they are not RNG-reproducible."""
    g = 8
    while g < int(n):
        g *= 2
    return g


class _GraphedDetector(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs_hallucinated, entry):
        if imgs_hallucinated.data_ptr() != entry.x.data_ptr():      # (the U-Net runner's static output is read in place)
            entry.x.copy_(imgs_hallucinated)
        entry.graph.replay()
        ctx.entry = entry
        return entry.total.clone()

    @staticmethod
    def backward(ctx, grad_total):
        e = ctx.entry
        # the graph differentiated loss * s with s = the loss scale at replay time.  fit_step seeds the backward pass with the scaler's own
        # scale tensor (LossScaler.backward): when that very tensor arrives and its value is the one the graph was replayed with, the
        # factor is exactly 1.0 and the graph's image gradient IS the result -- no launch, and the U-Net's backward graphs read it in
        # place (UnetRunner.adopt_static_dout).  Any other caller gets the chain rule.
        sc = e.scaler
        if sc is not None and sc.__dict__.get("_scale_t") is not None and grad_total.data_ptr() == sc._scale_t.data_ptr() \
                and sc._scale_t_value == e.scale_value:
            return e.dimg, None
        f = (grad_total * e.inv_scale).to(e.dimg.dtype)
        dst = e.unet_dout() if e.unet_dout is not None else None
        if dst is not None and dst.shape == e.dimg.shape and dst.dtype == e.dimg.dtype and dst.data_ptr() != e.dimg.data_ptr():
            return torch.mul(e.dimg, f, out=dst), None
        return e.dimg * f, None


class _Entry:
    pass


class DetectorStepGraph:
    def __init__(self, lit):
        self.lit = lit
        self.entries = {}                  # insertion order = recency (LRU: see step())
        self.max_entries = int(os.environ.get("HD_DET_GRAPH_MAX", "6"))
        self.pool = None
        self.usable = True
        self.replays = 0
        self.captures = 0
        self.fork_postprocess = os.environ.get("HD_DET_GRAPH_FORK", "0") == "1"      # A/B knob

    # -------------------------------------------------------------------------------------------------------- targets
    @staticmethod
    def _stage_table(lens, G, device):
        """[len(lens), G] gather rows into cat(boxes)+[zero row], and the live-row mask; cached per count tuple (pinned upload)."""
        key = ("staged", lens, G, str(device))
        hit = _PAD_IDX_CACHE.get(key)
        if hit is None:
            S, rows, live, lo = sum(lens), [], [], 0
            for n in lens:
                rows.append(list(range(lo, lo + n)) + [S] * (G - n))
                live.append([True] * n + [False] * (G - n))
                lo += n
            idx = torch.tensor(rows, dtype=torch.int64).reshape(-1).pin_memory().to(device, non_blocking=True)
            lv = torch.tensor(live, dtype=torch.bool).reshape(len(lens), G).pin_memory().to(device, non_blocking=True)
            if len(_PAD_IDX_CACHE) > 256:
                _PAD_IDX_CACHE.clear()
            hit = _PAD_IDX_CACHE[key] = (idx, lv)
        return hit

    def _stage_targets(self, e, targets):
        """Real target lists -> the entry's static [M, G, 4] / [M, G] buffers (one concatenation + one gather each)."""
        dev = e.tb.device
        # targets that are the very tensors staged last time, unmodified since (a resident batch re-used step after step), are not staged
        # again -- the rule the image batches follow (step()); the references held in e.tsrc keep their storage from being recycled
        sig = [(t["boxes"], t["boxes"]._version, t["labels"], t["labels"]._version) for t in targets]
        if e.tsrc is not None and len(e.tsrc) == len(sig) and all(a[0] is b[0] and a[1] == b[1] and a[2] is b[2] and a[3] == b[3] for a, b in zip(e.tsrc, sig)):
            return
        lens = tuple(int(t["boxes"].shape[0]) for t in targets)
        idx, live = self._stage_table(lens, e.G, dev)
        boxes = torch.cat([t["boxes"].to(torch.float32).reshape(-1, 4) for t in targets] + [e.zero_box], dim=0)
        labels = torch.cat([t["labels"].reshape(-1) for t in targets] + [e.zero_label], dim=0)
        torch.index_select(boxes, 0, idx, out=e.tb.view(-1, 4))
        torch.index_select(labels, 0, idx, out=e.tl.view(-1))
        e.live.copy_(live)
        # the reference's degenerate-box flag (eval_forward_fasterrcnn.py:41-53) of these rows, once per staging: the graph reads it
        e.deg.copy_(((e.tb[..., 2:] <= e.tb[..., :2]).any(dim=-1) & e.live).any().reshape(1))
        e.tsrc = sig

    # -------------------------------------------------------------------------------------------------------- capture
    def _section(self, e, x):
        lit, N = self.lit, e.N
        # rows [0, N) = IR targets (hallucinated pass), [N, 2N) = RGB, [2N, 3N) = IR again: the order the fused evaluation
        # concatenates the passes in, so that its stacks of these rows are views (detection.stack_rows), not copies
        t = [{"boxes": e.tb[i], "labels": e.tl[i], "_rows": e.live[i], "_flag": e.deg} for i in range(3 * N)]
        t_ir, t_rgb, t_ir2 = t[:N], t[N:2 * N], t[2 * N:]
        ir3 = e.ir.expand(-1, e.x.shape[1], -1, -1) if e.ir.shape[1] == 1 and e.x.shape[1] != 1 else e.ir
        return lit._detector_section(x, e.rgb, ir3, t_rgb, t_ir, 'train', False, targets_ir_pass=t_ir2)

    def _aliased_runner(self, imgs_hallucinated):
        """The U-Net runner when `imgs_hallucinated` IS its forward graph's static output buffer, else None."""
        runner = getattr(self.lit.encoder_decoder, "runner", None)
        rg = getattr(runner, "_g", None) if runner is not None and getattr(runner, "use_graphs", False) else None
        if rg is not None and rg.get("out") is not None and rg["out"].data_ptr() == imgs_hallucinated.data_ptr() \
                and rg["out"].shape == imgs_hallucinated.shape and imgs_hallucinated.is_contiguous():
            return runner
        return None

    def _build(self, key, imgs_hallucinated, imgs_rgb, imgs_ir, G):
        lit = self.lit
        dev = imgs_hallucinated.device
        e = _Entry()
        e.N, e.G = imgs_hallucinated.shape[0], G
        # the hallucinated batch: the U-Net runner's own static output buffer when it replays graphs (read in place: the entry is
        # keyed on its address), a private buffer otherwise
        runner = self._aliased_runner(imgs_hallucinated)
        if runner is not None:
            e.x = runner._g["out"].detach()
            e.unet_dout = lambda: (runner._g or {}).get("dout") if (runner._g or {}).get("bwd") is not None else None
        else:
            e.x = torch.empty_like(imgs_hallucinated, memory_format=torch.contiguous_format)
            e.unet_dout = None
        e.rgb = torch.empty_like(imgs_rgb, memory_format=torch.contiguous_format)
        e.src = [None, None, None, None]       # (rgb tensor, its version, ir tensor, its version) of the last staging
        # the IR batch may arrive as the stride-0 three-channel view of a one-channel batch: keep one plane, expand inside
        one_plane = imgs_ir.dim() == 4 and imgs_ir.shape[1] > 1 and imgs_ir.stride(1) == 0
        e.ir_one_plane = one_plane
        e.ir = torch.empty((imgs_ir.shape[0], 1) + tuple(imgs_ir.shape[2:]), dtype=imgs_ir.dtype, device=dev) if one_plane \
            else torch.empty_like(imgs_ir, memory_format=torch.contiguous_format)
        e.tb = torch.zeros((3 * e.N, G, 4), dtype=torch.float32, device=dev)
        e.tl = torch.zeros((3 * e.N, G), dtype=torch.int64, device=dev)
        e.live = torch.zeros((3 * e.N, G), dtype=torch.bool, device=dev)
        e.deg = torch.zeros((1,), dtype=torch.uint8, device=dev)
        e.tsrc = None
        e.zero_box = torch.zeros((1, 4), dtype=torch.float32, device=dev)
        e.zero_label = torch.zeros((1,), dtype=torch.int64, device=dev)
        e.scale = torch.ones((), dtype=torch.float32, device=dev)
        e.inv_scale = torch.ones((), dtype=torch.float32, device=dev)
        e.scale_value = None
        e.scaler = lit.scaler
        e.runner = runner
        return e

    def _capture(self, e):
        """Warm up eagerly on the staged inputs (lazy caches: anchors, workspaces, index tables), then capture."""
        def body():
            # a fresh leaf over the static buffer in every run: its gradient sink is then created on the stream of THIS run (a leaf
            # kept from the warm-up would tie the captured backward pass to the warm-up's stream, which capture cannot wait on)
            x = e.x.detach().requires_grad_(True)
            losses_det, total, dets = self._section(e, x)
            # optional fork (HD_DET_GRAPH_FORK=1): the three passes' post-processing (score filter, top-k, decode, NMS: ~25 small,
            # latency-bound launches that depend on the forward pass only) on a second branch of the graph, concurrent with the
            # backward pass.  Measured and left OFF: 13.27-13.29 ms per step with the fork vs 13.07-13.09 without (same box,
            # alternating runs) -- a two-branch hipGraph costs more in cross-branch synchronisation than the overlap returns.
            cur = torch.cuda.current_stream()
            branch = torch.cuda.Stream() if self.fork_postprocess else cur
            branch.wait_stream(cur)
            with torch.cuda.stream(branch):
                for d in dets:
                    if hasattr(d, "flush"):
                        d.flush()
            (dimg,) = torch.autograd.grad(total, x, grad_outputs=e.scale)      # = d(total * scale)/dx without the extra launches
            cur.wait_stream(branch)
            return losses_det, total, dets, dimg

        _eff._GRAPH_FLAGS = []
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            _eff._GRAPH_FLAGS = flags = []
            g = torch.cuda.CUDAGraph()
            kw = dict(pool=self.pool) if self.pool is not None else {}
            with capture_without_gc(), torch.cuda.graph(g, capture_error_mode="thread_local", **kw):
                losses_det, total, dets, dimg = body()
        finally:
            _eff._GRAPH_FLAGS = None
        if self.pool is None:
            self.pool = g.pool()
        e.graph, e.total, e.dimg = g, total, dimg
        if e.runner is not None:
            e.runner.adopt_static_dout(dimg)      # the U-Net's backward graphs may take this buffer as their static input
        e.losses = {k: v for k, v in losses_det.items()}
        e.flag = flags[0] if flags else None
        e.det_pads = []
        for d in dets:
            if not isinstance(d, LazyDetections) or d._pad is None:
                raise RuntimeError("detector graph: detections are expected as flushed LazyDetections")
            e.det_pads.append(d._pad)
        self.captures += 1

    # -------------------------------------------------------------------------------------------------------- per step
    def step(self, imgs_hallucinated, imgs_rgb, imgs_ir, targets_rgb, targets_ir):
        lit = self.lit
        N = imgs_hallucinated.shape[0]
        targets = list(targets_ir) + list(targets_rgb) + list(targets_ir)
        for t in targets:
            _eff._check_targets([t])
        G = _bucket(max([1] + [int(t["boxes"].shape[0]) for t in targets]))
        w = tuple(float(lit_w) for lit_w in self._weights())
        alias = imgs_hallucinated.data_ptr() if self._aliased_runner(imgs_hallucinated) is not None else 0
        key = (tuple(imgs_hallucinated.shape), alias, tuple(imgs_rgb.shape), tuple(imgs_ir.shape),
               imgs_ir.stride(1) == 0, G, w, lit.detector_name, bool(lit.detector.transform.training))
        e = self.entries.get(key)
        # the loss scale of this step: GradScaler order (the overflow check of step t decides the scale of step t+1), resolved
        # BEFORE the replay because the scaled backward pass is inside the graph
        lit.scaler.resolve()
        s = float(lit.scaler.scale_value)
        fresh = e is None
        if fresh:
            e = self._build(key, imgs_hallucinated, imgs_rgb, imgs_ir, G)
        if e.scale_value != s:
            e.scale.fill_(s)
            e.inv_scale.fill_(1.0 / s)
            e.scale_value = s
        # a batch that is the very tensor staged last time, unmodified since (a resident batch re-used step after step), is not
        # copied again; the reference held in e.src keeps its storage from being recycled under that test
        if not (e.src[0] is imgs_rgb and e.src[1] == imgs_rgb._version):
            e.rgb.copy_(imgs_rgb)
            e.src[0], e.src[1] = imgs_rgb, imgs_rgb._version
        ir_base = imgs_ir._base if imgs_ir._base is not None else imgs_ir
        if not (e.src[2] is ir_base and e.src[3] == ir_base._version):
            e.ir.copy_(imgs_ir[:, :1] if e.ir_one_plane else imgs_ir)
            e.src[2], e.src[3] = ir_base, ir_base._version
        self._stage_targets(e, targets)
        if fresh:
            if e.x.data_ptr() != imgs_hallucinated.data_ptr():
                e.x.copy_(imgs_hallucinated.detach())
            try:
                self._capture(e)
            except Exception as err:                   # loud, then the eager path for the rest of the run
                self.usable = False
                warnings.warn("hallucidet_amd: detector hipGraph capture failed (%s: %s); the detector half stays eagerly issued"
                              % (type(err).__name__, err))
                return lit._detector_section(imgs_hallucinated, imgs_rgb, imgs_ir, targets_rgb, targets_ir, 'train', False)
            self.entries[key] = e
            # one captured ~700-kernel graph per (shapes, G bucket, weights, ...), each pinning its static buffers in the shared pool:
            # keep the most recently used few (a dataset with a wide box-count spread visits many G buckets)
            while len(self.entries) > self.max_entries:
                self.entries.pop(next(iter(self.entries)))
        else:
            self.entries[key] = self.entries.pop(key)          # most recently used last
        # the flag the PREVIOUS batch left behind is read now (a bad box raises the reference's assertion one call later)
        pend = lit.detector.__dict__.pop("_pending_degenerate", None)
        if pend is not None:
            pend.raise_if_set()
        total = _GraphedDetector.apply(imgs_hallucinated, e)
        self.replays += 1
        if e.flag is not None:
            lit.detector.__dict__["_pending_degenerate"] = _eff._AsyncFlag(e.flag, targets)
        dets = tuple(LazyDetections(*p) for p in e.det_pads)
        losses = dict(e.losses)
        return losses, total, dets

    def _weights(self):
        from .config import Config
        w = Config.Losses.hparams_losses_weights
        return [w[wk] for _, wk in self.lit._loss_keys()]
