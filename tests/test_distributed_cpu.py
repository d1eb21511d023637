"""World-size-2 gloo tests of the data-parallel exchange (hallucidet_amd/distributed.py): the same code path RCCL runs on
the GPUs.  Checks DDP semantics: mean of per-rank gradients, bucketed slices cover the arena exactly, start-up broadcast."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hallucidet_amd.distributed import GradientAverager, broadcast_parameters, is_dist
    assert is_dist()
    g = torch.Generator().manual_seed(100 + rank)
    grads = torch.randn(n, generator=g)
    params = torch.full((n,), float(rank))
    buf = torch.full((7,), float(rank + 10))
    broadcast_parameters(params, [buf])
    av = GradientAverager(n_buckets=4)
    av.start(grads)
    av.finish(grads)
    q.put((rank, grads.numpy().copy(), params.numpy().copy(), buf.numpy().copy()))   # numpy: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10007, 24436659 // 64])
def test_gradient_mean_and_broadcast_world2(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, g, p, b = q.get(timeout=120)
        res[r] = (torch.from_numpy(g), torch.from_numpy(p), torch.from_numpy(b))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for r in range(world):
        g, p, b = res[r]
        assert torch.allclose(g, want, atol=1e-6), "rank %d gradient is not the mean over ranks" % r
        assert torch.equal(p, torch.zeros(n)) and torch.equal(b, torch.full((7,), 10.0)), "rank 0 state was not broadcast"


def _model_worker(rank, world, port, inject_inf, q):
    """One data-parallel training step of an EncoderDecoderLit-shaped object on CPU: the REAL ParamArena / LossScaler /
    GradientAverager / broadcast_parameters / exchange_and_step, bucket hooks fired in backward-completion order, and a torch
    restatement of the fused Adam kernel (hd_adam_step is GPU-only) with the same skip-on-overflow contract."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn as nn
    from hallucidet_amd.distributed import GradientAverager, broadcast_parameters, exchange_and_step
    from hallucidet_amd.optim import FusedAdam, LossScaler, ParamArena

    class CpuAdam(FusedAdam):
        def step(self, closure=None, inv_scale=1.0, check_inf=False):
            r, g_ = self.runner, self.param_groups[0]
            grads = r.flat_grads
            self.found_inf.fill_(0.0 if bool(torch.isfinite(grads).all()) else 1.0)
            self.step_count += 1
            if not (check_inf and float(self.found_inf) != 0.0):
                gg = (grads * inv_scale).clamp(-g_["clip_value"], g_["clip_value"])       # hd_adam_step: g * inv_scale first
                b1, b2 = g_["betas"]
                self.exp_avg.mul_(b1).add_(gg, alpha=1 - b1)
                self.exp_avg_sq.mul_(b2).addcmul_(gg, gg, value=1 - b2)
                mh = self.exp_avg / (1 - b1 ** self.step_count)
                vh = self.exp_avg_sq / (1 - b2 ** self.step_count)
                r.flat_params.addcdiv_(mh, vh.sqrt().add_(g_["eps"]), value=-g_["lr"])
            self.last_inv_scale = inv_scale
            self._inf_host.copy_(self.found_inf)
            self._inf_pending = True

    torch.manual_seed(7 + rank)                       # every rank starts from DIFFERENT weights: the broadcast must fix that
    net = nn.Sequential(nn.Conv2d(1, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(),
                        nn.Conv2d(4, 3, 3, padding=1))
    arena = ParamArena(net.parameters())
    bufs = [b for b in net.buffers() if b.dtype.is_floating_point]
    bufs[0].fill_(float(rank + 1))
    broadcast_parameters(arena.flat_params, bufs)
    start, buf_start = arena.flat_params.clone(), bufs[0].clone()      # (running statistics then evolve per rank: no SyncBN in the reference)
    opt = CpuAdam(arena, lr=1e-2, clip_value=0.5)
    scaler = LossScaler(arena, init_scale=1024.0)
    av = GradientAverager()
    x = torch.rand(2, 1, 8, 8, generator=torch.Generator().manual_seed(50 + rank))         # per-rank shard
    loss = net(x).square().mean()
    g = arena.flat_grads
    g.zero_()
    av.begin(g)
    scaler.scale(loss).backward()
    g.div_(scaler.scale_value)                         # the HIP kernels emit parameter gradients already divided by the scale
    own = g.clone()
    if inject_inf and rank == 1:
        g[3] = float("inf")
    n = g.numel()
    cuts = [n, n * 3 // 4 // 4 * 4, n // 2 // 4 * 4, 16, 0]      # five buckets, reported from the END of the arena (as backward does)
    for hi, lo in zip(cuts[:-2], cuts[1:-1]):
        av.bucket_ready(lo, hi)                        # the last slice [0, 16) is left to start(): no hook may be assumed
    issued_by_hooks = list(av.issued)
    exchange_and_step(av, g, scaler, opt)
    skipped = scaler.resolve()
    q.put((rank, own.numpy().copy(), g.numpy().copy(), start.numpy().copy(), arena.flat_params.detach().numpy().copy(), issued_by_hooks,
           list(av.issued), bool(skipped), scaler.scale_value, opt.step_count, buf_start.numpy().copy(), opt.last_inv_scale,
           opt.exp_avg.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,inject_inf", [(2, False), (2, True), (4, False), (4, True)])
def test_model_level_data_parallel_step(world, inject_inf):
    """World sizes 2 and 4: the bucket cover, the start-up broadcast and the identical end state do not depend on the world size;
    the mean does (x 1 / world, applied by the optimizer: exchange_and_step leaves the SUM in the arena)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, inject_inf, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=180)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    own = [res[r][0] for r in range(world)]
    summed0, start0, after0, hooks0, issued0 = res[0][1:6]
    n = own[0].size
    for r in range(1, world):
        _, _, start_r, _, hooks_r, _, _, _, _, buf_r = res[r][:10]
        assert (start0 == start_r).all() and (res[0][9] == buf_r).all(), "ranks must start from rank 0's parameters and BatchNorm buffers"
        assert hooks0 == hooks_r
    # buckets: hooks fire from the end of the arena, start() adds what they left, together exactly one cover of [0, n)
    assert [hi for _, hi in hooks0] == sorted([hi for _, hi in hooks0], reverse=True) and hooks0[0][1] == n
    cover = sorted(issued0)
    assert cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    for r in range(world):
        summed, _, after, _, _, skip, scale, steps, _, inv, m1 = res[r][1:12]
        assert inv == 1.0 / world, "the optimizer must receive 1 / world as its inverse scale"
        if not inject_inf:
            assert np.allclose(summed, sum(own), rtol=1e-6, atol=1e-9) and (summed == summed0).all(), "exchanged gradient != sum of per-rank gradients"
            # what Adam consumed is the MEAN: its first moment after one step is (1 - beta1) * clip(mean)
            assert np.allclose(m1, 0.1 * np.clip(sum(own) / world, -0.5, 0.5), rtol=1e-5, atol=1e-9), "the optimizer did not see the mean gradient"
            assert not skip and steps == 1
            assert (after == after0).all() and not (after == start0).all(), "ranks must hold identical, updated parameters after the step"
        else:
            assert skip, "an overflow on one rank must skip the step on every rank"
            assert (after == start0).all() and steps == 0
            assert scale == 512.0


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from hallucidet_amd.distributed import GradientAverager, is_dist
    assert not is_dist()
    g = torch.arange(5.0)
    av = GradientAverager()
    av.start(g)
    av.finish(g)
    assert torch.equal(g, torch.arange(5.0))


def test_self_launcher_starts_one_rank_per_gpu_and_relays_rank0(tmp_path):
    """`python bench.py --gpus N` without torchrun: hallucidet_amd.launch re-runs the command N times with the torch.distributed.run
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR 127.0.0.1 / a free MASTER_PORT) and relays rank 0's output.  Driven
    here with a two-rank gloo child (what bench.py does with RCCL): the ranks rendezvous, all-reduce, rank 0's JSON line comes
    back through the launcher; a failing rank's exit code is the launcher's and the surviving rank is not left behind."""
    import io
    import json
    import sys
    from hallucidet_amd import launch
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys, time\n"
        "import torch, torch.distributed as dist\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "if len(sys.argv) > 1 and sys.argv[1] == 'fail' and rank == 1:\n"
        "    sys.exit(3)\n"
        "dist.init_process_group('gloo', rank=rank, world_size=world)\n"
        "t = torch.tensor([float(rank + 1)])\n"
        "dist.all_reduce(t)\n"
        "print('noise from rank %d' % rank) if rank else print(json.dumps({'sum': float(t), 'rccl_ranks': dist.get_world_size()}), flush=True)\n"
        "dist.barrier()\n"
        "dist.destroy_process_group()\n")
    assert launch.need_self_launch(1) is False
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    buf = io.StringIO()
    rc = launch.launch_ranks(2, [sys.executable, str(child)], env=env, timeout=120, out=buf)
    assert rc == 0, buf.getvalue()
    lines = [l for l in buf.getvalue().splitlines() if l.strip() and not l.startswith("[Gloo]")]      # gloo's own connection banner
    assert len(lines) == 1 and json.loads(lines[0]) == {"sum": 3.0, "rccl_ranks": 2}      # rank 1's stdout is not relayed
    assert "noise from rank 1" not in buf.getvalue()
    buf = io.StringIO()
    rc = launch.launch_ranks(2, [sys.executable, str(child), "fail"], env=env, timeout=120, out=buf)
    assert rc == 3


def test_bench_rank_validation_through_the_launcher(tmp_path):
    """bench.py's own N > 1 entry, as far as a box without a GPU can take it: (1) `python bench.py --gpus 2` started plainly
    becomes the launcher (hallucidet_amd.launch) -- two children with RANK / LOCAL_RANK / WORLD_SIZE, whose failure ("no GPU
    visible": there is no CPU path) is the launcher's exit code; (2) a WORLD_SIZE that contradicts --gpus and (3) a RANK outside
    the world are refused BEFORE anything touches a device, with a message that says how to launch."""
    import io
    import subprocess
    import sys
    from hallucidet_amd import launch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HD_FORCE_DIST")}
    env["HIP_VISIBLE_DEVICES"] = ""          # the same on a GPU box: no device for the children
    env["CUDA_VISIBLE_DEVICES"] = ""
    # (1) plain start: the parent launches its ranks and relays the first failing exit code
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stderr.count("no GPU visible") >= 1, r.stderr[-2000:]
    assert r.stdout.strip() == ""            # no JSON line from a run that did not happen
    # (2) two ranks, but the command says --gpus 3
    buf = io.StringIO()
    rc = launch.launch_ranks(2, [sys.executable, bench, "--gpus", "3"], env=env, timeout=300, out=buf)
    assert rc != 0 and buf.getvalue().strip() == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "3"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "started with WORLD_SIZE=2" in r.stderr and "--nproc-per-node 3" in r.stderr
    # (3) a rank outside the world
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="2", RANK="2", LOCAL_RANK="2"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "outside WORLD_SIZE=2" in r.stderr


def test_loss_scaler_backward_seeds_with_the_scale():
    """LossScaler.backward(loss) == scale(loss).backward(): the backward pass seeded with the scale itself (round 6: no loss * scale,
    ones_like, MulBackward launches); the persistent scale tensor follows the scale value."""
    from hallucidet_amd.optim import LossScaler, ParamArena
    torch.manual_seed(0)
    net = torch.nn.Linear(5, 3)
    arena = ParamArena(net.parameters())
    scaler = LossScaler(arena, init_scale=512.0)
    x = torch.randn(4, 5)
    loss = net(x).square().mean()
    scaler.scale(loss).backward()
    want = arena._gflat.clone()
    arena._gflat.zero_()
    loss = net(x).square().mean()
    scaler.backward(loss)
    assert torch.equal(arena._gflat, want) and arena.grad_scale == 512.0
    t = scaler.scale_tensor(loss)
    assert float(t) == 512.0 and scaler.scale_tensor(loss) is t
    scaler.scale_value = 256.0
    assert float(scaler.scale_tensor(loss)) == 256.0
