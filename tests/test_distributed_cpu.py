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


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from hallucidet_amd.distributed import GradientAverager, is_dist
    assert not is_dist()
    g = torch.arange(5.0)
    av = GradientAverager()
    av.start(g)
    av.finish(g)
    assert torch.equal(g, torch.arange(5.0))
