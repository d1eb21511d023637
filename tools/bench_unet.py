"""Time the hallucination-net forward+backward+Adam at the BASELINE geometry (8 x 3 x 512 x 640)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd.models.encoder_decoder import EncoderDecoder
from hallucidet_amd.optim import FusedAdam

N = int(os.environ.get("N", 8)); H, W = 512, 640
steps = int(os.environ.get("STEPS", 5))
dev = torch.device("cuda:0")
torch.manual_seed(123)
net = EncoderDecoder(name="resnet34").encoder_decoder.to(dev).train()
opt = FusedAdam(net, lr=1e-4, clip_value=0.5)
net.runner.enable_graphs(bool(int(os.environ.get('HD_GRAPHS', '0'))))
x = torch.rand(N, 3, H, W, device=dev)
g = torch.randn(N, 3, H, W, device=dev) * 1e-3
net.runner.grad_scale = 1024.0
def step():
    out = net(x)
    out.backward(g * 1024.0)
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("unet fwd+bwd+adam: %.2f ms/step  -> %.1f img/s ; %.1f TFLOP/s (233.4 GFLOP/img)" % (dt * 1e3, N / dt, 233.4e9 * N / dt / 1e12))
# forward only
torch.cuda.synchronize(); t0 = time.time()
with torch.no_grad():
    for _ in range(steps): net(x)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("unet fwd (train-mode BN, no save): %.2f ms -> %.1f TFLOP/s (78.33 GFLOP/img)" % (dt * 1e3, 78.33e9 * N / dt / 1e12))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
