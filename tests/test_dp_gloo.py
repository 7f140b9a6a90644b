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


def _wire_and_algo(rank, world):
    """GradBuckets(wire=, algo=): every rank ends with IDENTICAL gradients (replicas cannot drift), the fp32 rs_ag form equals the
    all-reduce up to summation order, the bf16 wire within bf16 rounding of it."""
    from hsenet_amd.dp import GradBuckets
    torch.manual_seed(300 + rank)
    base = torch.randn(1003) * torch.logspace(-3, 1, 1003)          # (1003: not divisible by the world size -> the padded rs_ag path)
    all_ = [torch.empty(1003) for _ in range(world)]
    dist.all_gather(all_, base)
    ref = sum(a.double() for a in all_)
    bounds = [(700, 1003), (256, 700), (0, 256)]
    worst = {}
    for wire in ("fp32", "bf16"):
        for algo in ("allreduce", "rs_ag", "auto"):
            flat = base.clone()
            gb = GradBuckets(flat, bounds, wire=wire, algo=algo, rs_ag_min_bytes=1024)
            for i in range(3):
                gb.reduce(i)
            gb.wait()
            others = [torch.empty(1003) for _ in range(world)]
            dist.all_gather(others, flat)
            assert all(torch.equal(o, flat) for o in others), (wire, algo)           # replica-identical
            err = float((flat.double() - ref).norm() / ref.norm())
            worst[(wire, algo)] = err
            # (a bf16 SUM over the ranks rounds after every addition: ~2e-3 at 2 ranks, 4.1e-3 at the reference's 8)
            assert err < (1e-6 if wire == "fp32" else 1e-3 + 2e-3 * world ** 0.5), (wire, algo, err)
    if rank == 0:
        print("GradBuckets relative error vs the exact sum:", {f"{k[0]}/{k[1]}": f"{v:.1e}" for k, v in worst.items()})
    # auto at two ranks or for small buckets falls back to one all-reduce; a muted instance launches nothing
    gb = GradBuckets(base.clone(), bounds, algo="auto", rs_ag_min_bytes=1 << 30)
    assert not any(p[0] for p in gb._plan)
    gb.muted = True
    gb.reduce(0)
    gb.wait()
    try:
        GradBuckets(base.clone(), bounds, wire="fp16")
        raise AssertionError("unknown wire format accepted")
    except ValueError:
        pass


def test_grad_buckets_allreduce_gloo():
    _run(_buckets)


@pytest.mark.parametrize("world", [2, 4, 8])      # 8 = the reference's node (README.md:78-105): 1003 elements pad to 1008 for rs_ag
def test_grad_buckets_wire_formats_and_algorithms_gloo(world):
    _run(_wire_and_algo, world)


def test_gather_features_world8_gloo():
    """C2 at the reference's world size: one fused all-gather of 8 x [3, 16] features, backward = this rank's slice of the summed gradient."""
    _run(_gather, 8)


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
