import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hallucidet_amd import synthetic
from hallucidet_amd.det_graph import DetectorStepGraph
from hallucidet_amd.utils import eval_forward_fasterrcnn as eff
stage = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "fasterrcnn"
lit = synthetic.make_module(seed=5, device="cuda", detector_name=name)
b = synthetic.make_batch(2, 128, 160, seed=9, device="cuda")
lit.encoder_decoder.train()
ir3 = b[2].expand(-1, 3, -1, -1)
hall = lit.encoder_decoder(ir3).detach()
G = DetectorStepGraph(lit)
e = G._build(None, hall, b[0], ir3, 8)
e.scale.fill_(65536.0)
e.rgb.copy_(b[0]); e.ir.copy_(b[2]); G._stage_targets(e, list(b[3]) + list(b[1]))
e.x.copy_(hall)
det = lit.detector

def body():
    if stage == "transform":
        il, _ = det.transform.forward_batches([e.x, e.rgb, e.ir.expand(-1, 3, -1, -1)], None)
        return il.tensors
    if stage == "backbone":
        il, _ = det.transform.forward_batches([e.x, e.rgb, e.ir.expand(-1, 3, -1, -1)], None)
        f = det.backbone(il.tensors, n_active=2)
        return list(f.values())
    if stage == "backbone_bwd":
        il, _ = det.transform.forward_batches([e.x, e.rgb, e.ir.expand(-1, 3, -1, -1)], None)
        f = det.backbone(il.tensors, n_active=2)
        s = sum((getattr(v, "_hd_active", v)).float().sum() for v in f.values())
        return torch.autograd.grad(s, e.x)
    x = e.x.detach().requires_grad_(True)
    if stage == "section":
        return G._section(e, x)[1]
    if stage == "grad":
        l, t, d = G._section(e, x)
        return torch.autograd.grad(t * e.scale, x)
    if stage == "full":
        l, t, d = G._section(e, x)
        g = torch.autograd.grad(t * e.scale, x)
        for x in d: x.flush()
        return g
eff._GRAPH_FLAGS = []
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    out = body()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", stage, flush=True)
