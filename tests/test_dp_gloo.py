"""world_size-2 CPU (gloo) tests of the data-parallel exchange (hsenet_amd/dp.py): the same code path that runs
over RCCL/xGMI with one process per GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn), nprocs=world, join=True)


def _buckets(rank, world):
    from hsenet_amd.dp import GradBuckets
    torch.manual_seed(100 + rank)
    flat = torch.randn(1000)
    mine = flat.clone()
    all_ = [torch.empty(1000) for _ in range(world)]
    dist.all_gather(all_, mine)
    gb = GradBuckets(flat, [(600, 1000), (250, 600), (0, 250)])
    assert gb.world == world and abs(gb.grad_scale - 1.0 / world) < 1e-12
    for i in range(3):               # buckets are launched in backward-completion order, asynchronously
        gb.reduce(i)
    gb.wait()
    assert torch.allclose(flat, sum(all_), atol=1e-6)
    assert torch.allclose(flat * gb.grad_scale, sum(all_) / world, atol=1e-6)
    for bad in ([(0, 2000)], [(0, 400), (500, 1000)], [(0, 600), (500, 1000)], [(0, 900)]):
        try:                         # out of range / gap / overlap / short: the buckets must partition the flat gradient
            GradBuckets(flat, bad)
            raise AssertionError(f"bad bucket plan accepted: {bad}")
        except ValueError:
            pass


def _gather(rank, world):
    from hsenet_amd.dp import gather_features
    torch.manual_seed(rank)
    img = torch.randn(3, 8, requires_grad=True)
    txt = torch.randn(3, 8, requires_grad=True)
    ai, at = gather_features(img, txt)
    assert ai.shape == (3 * world, 8) and at.shape == (3 * world, 8)
    assert torch.equal(ai[rank * 3:(rank + 1) * 3], img.detach()) and torch.equal(at[rank * 3:(rank + 1) * 3], txt.detach())
    # loss couples every rank's rows: d/d(img_local) must include the other ranks' contributions (gather_with_grad)
    w = torch.arange(3 * world * 8, dtype=torch.float32).view(3 * world, 8) / 100.0
    loss = (ai * w).sum() * (rank + 1) + (at * w).sum()
    loss.backward()
    coef = sum(r + 1 for r in range(world))
    assert torch.allclose(img.grad, coef * w[rank * 3:(rank + 1) * 3], atol=1e-5)
    assert torch.allclose(txt.grad, world * w[rank * 3:(rank + 1) * 3], atol=1e-5)


def test_grad_buckets_allreduce_gloo():
    _run(_buckets)


def test_gather_features_gloo():
    _run(_gather)


def test_world1_is_identity():
    from hsenet_amd.dp import GradBuckets, gather_features
    g = torch.ones(10)
    gb = GradBuckets(g, [(0, 10)])
    gb.reduce(0)
    gb.wait()
    assert gb.world == 1 and gb.grad_scale == 1.0 and torch.equal(g, torch.ones(10))
    a, b = torch.randn(2, 4), torch.randn(2, 4)
    ga, gbb = gather_features(a, b)
    assert ga is a and gbb is b
