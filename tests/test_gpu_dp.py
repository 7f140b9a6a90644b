"""DP equivalence on the GPU box: 2 ranks x 1 volume (flat-gradient buckets all-reduced while backward runs, mean folded
into AdamW) must produce the same parameters as 1 rank x 2 volumes.  Both ranks share cuda:0 and talk over gloo — the
control flow is the one bench.py runs over RCCL with one process per GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GEOM = dict(in_channels=1, img_size=(8, 64, 64), patch_size=(4, 16, 16), num_layers=4, pos_embed="perceptron", classification=True)


def _model(dev):
    import hsenet_amd as H
    from oracle import ref_cpu as R
    m = H.ViT_stage1(**GEOM)
    m.load_state_dict(R.seeded_state_dict(R.ViT_stage1(**GEOM), 77))
    return m.to(dev)


def _vols():
    from oracle import ref_cpu as R
    return R.synthetic_volume((2, 1, 8, 64, 64), 78)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsenet_amd.engine import FlatEncoder
        dev = torch.device("cuda:0")
        eng = FlatEncoder(_model(dev), 1, lr=1e-3, weight_decay=0.1, n_buckets=3)
        assert eng.buckets.world == 2 and len(eng.layer_chunks) == 2
        x = _vols()[rank:rank + 1].to(dev)
        for _ in range(2):
            eng.train_step(x)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save(eng.flat.cpu(), out)
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_single_rank(dev, tmp_path):
    from hsenet_amd.engine import FlatEncoder
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp2.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    dp2 = torch.load(out)
    eng = FlatEncoder(_model(dev), 2, lr=1e-3, weight_decay=0.1)
    x = _vols().to(dev)
    p0 = eng.flat.clone()
    for _ in range(2):
        eng.train_step(x)
    torch.cuda.synchronize()
    single = eng.flat.cpu()
    moved = float((single - p0.cpu()).norm())
    diff = float((single - dp2).norm())
    print(f"param movement {moved:.3e}, |dp2 - single| {diff:.3e}")
    # identical math up to bf16 rounding of per-rank vs batched activations and fp32 atomic summation order
    assert moved > 0 and diff < 2e-2 * moved


def test_flat_encoder_matches_autograd_path(dev):
    """the engine's fused fwd+bwd (flat buffers) equals the module/autograd path on the same weights and input."""
    from hsenet_amd.engine import FlatEncoder
    m_ref, m_eng = _model(dev), _model(dev)
    x = _vols().to(dev)
    y, _ = m_ref(x)
    (y * y).mean().backward()
    eng = FlatEncoder(m_eng, 2)
    ye = eng.forward(x)
    from hsenet_amd import _lib as L, functional as Fn
    L.call("hs_scale", ye.data_ptr(), eng.dy.data_ptr(), 2.0 / ye.numel(), ye.numel(), Fn._s())
    eng.backward(eng.dy)
    torch.cuda.synchronize()
    assert float((ye - y.detach()).abs().max()) < 1e-5
    for (n, p), (_, q) in zip(m_eng.named_parameters(), m_ref.named_parameters()):
        e = float((p.grad - q.grad).norm() / q.grad.norm().clamp_min(1e-20))
        assert e < 2e-3, (n, e)       # atomics order / bucket chunking only


# ---- the stage-2 CLIP step (configs[3]) across ranks: all-gather of the cls features inside the step + bucketed all-reduce ----
CLIP_GEOM = dict(img_size=(8, 64, 64), patch_size=(4, 16, 16), num_layers=2)


def _clip_model(dev):
    from hsenet_amd.clip import M3DCLIPVision_stage2
    from oracle import clip_cpu as RC, ref_cpu as R
    ref = RC.M3DCLIPVision_stage2(CLIP_GEOM["img_size"], CLIP_GEOM["patch_size"], CLIP_GEOM["num_layers"])
    sd = R.seeded_state_dict(ref, 87)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)))
    sd["stage1_pretrained_CLIP.logit_scale"] = sd["logit_scale"].clone()
    m = M3DCLIPVision_stage2(**CLIP_GEOM)
    m.load_state_dict(sd)
    return m.to(dev).eval()


def _clip_inputs(n):
    from oracle import clip_cpu as RC, ref_cpu as R
    return (R.synthetic_volume((n, 1, 8, 64, 64), 88), R.synthetic_slices(n, 89), RC.synthetic_text_features(n, 90),
            RC.synthetic_text_features(n, 91))


def _clip_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsenet_amd.clip_step import ClipStage2Engine
        from hsenet_amd import clip
        dev = torch.device("cuda:0")
        eng = ClipStage2Engine(_clip_model(dev), 4, lr=1e-3)
        assert eng.world == 2 and eng.all.shape[0] == 8
        x, s, t1, t2 = (t[rank * 4:(rank + 1) * 4].to(dev) for t in _clip_inputs(8))
        for _ in range(2):
            eng.train_step(x, s, t1, t2)
        torch.cuda.synchronize()
        # local_loss branch across ranks: sum over ranks of the per-rank losses, gradient w.r.t. this rank's features
        im, tm = t1.clone().requires_grad_(), t2.clone().requires_grad_()
        sc = torch.tensor(2.0, device=dev)
        loss, lpi, lpt = clip.image_text_contrastive_learning(im, tm, sc, local_loss=True)
        loss.backward()
        torch.cuda.synchronize()
        torch.save(dict(flat=eng.flat.cpu(), loss=float(loss), gi=im.grad.cpu(), gt=tm.grad.cpu(), lpi=lpi.detach().cpu()), f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


def test_clip_step_two_ranks_match_single_rank(dev, tmp_path):
    """2 ranks x 4 volumes == 1 rank x 8 volumes for the stage-2 CLIP step (the cls features are all-gathered inside the step, the
    logits are 8x8 on both ranks); and the local_loss=True branch across 2 ranks vs the oracle's restatement of each rank's loss."""
    from hsenet_amd.clip_step import ClipStage2Engine
    from oracle import clip_cpu as RC
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "clip2")
    mp.spawn(_clip_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["flat"], r1["flat"])                      # replicas stay identical
    eng = ClipStage2Engine(_clip_model(dev), 8, lr=1e-3)
    x, sl, t1, t2 = (t.to(dev) for t in _clip_inputs(8))
    p0 = eng.flat.clone()
    for _ in range(2):
        eng.train_step(x, sl, t1, t2)
    torch.cuda.synchronize()
    single = eng.flat.cpu()
    moved, diff = float((single - p0.cpu()).norm()), float((single - r0["flat"]).norm())
    print(f"clip step: param movement {moved:.3e}, |dp2 - single| {diff:.3e}")
    assert moved > 0 and diff < 3e-2 * moved
    # local_loss: rank r's loss uses its 4 rows against all 8; gather_with_grad sums every rank's gradient into the local rows
    _, _, t1c, t2c = _clip_inputs(8)
    ir, tr = t1c.clone().requires_grad_(), t2c.clone().requires_grad_()
    total = 0
    for r, got in enumerate((r0, r1)):
        lo, lpi, _ = RC.image_text_contrastive_learning(ir[r * 4:(r + 1) * 4], tr[r * 4:(r + 1) * 4], torch.tensor(2.0), local_loss=True,
                                                        world=2, rank=r, all_image_features=ir, all_text_features=tr)
        assert abs(float(lo) - got["loss"]) < 1e-5 and float((lpi.detach() - got["lpi"]).abs().max()) < 1e-5
        total = total + lo
    total.backward()
    for r, got in enumerate((r0, r1)):
        assert float((got["gi"] - ir.grad[r * 4:(r + 1) * 4]).abs().max()) < 1e-5
        assert float((got["gt"] - tr.grad[r * 4:(r + 1) * 4]).abs().max()) < 1e-5


def _nccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)            # RCCL, as bench.py initialises it
    try:
        from hsenet_amd.engine import FlatEncoder
        from hsenet_amd.dp import _AllGatherCat
        eng = FlatEncoder(_model(dev), 2, lr=1e-3, weight_decay=0.1, n_buckets=2)
        eng.buckets.always = True                                                      # run the collectives at world size 1
        x = _vols().to(dev)
        for _ in range(2):
            eng.train_step(x)
        torch.cuda.synchronize()
        a = torch.randn(3, 8, device=dev, requires_grad=True)
        g = _AllGatherCat.apply(a, None)                                                # all_gather_into_tensor + all_reduce backward on RCCL
        g.sum().backward()
        torch.cuda.synchronize()
        torch.save(dict(flat=eng.flat.cpu(), ok=bool(torch.equal(g.detach(), a.detach()) and torch.equal(a.grad, torch.ones_like(a)))), out)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_world1_matches_no_process_group(dev, tmp_path):
    """the RCCL ("nccl") code path on the one GPU this box has: process group with device_id as bench.py creates it, the engine's
    asynchronous bucket all-reduces FORCED on at world size 1 (identity collectives, launched while the backward runs), the
    gathered-feature collectives; the trained parameters must equal those of a run without any process group."""
    from hsenet_amd.engine import FlatEncoder
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "nccl1.pt")
    mp.spawn(_nccl_worker, args=(1, port, out), nprocs=1, join=True)
    got = torch.load(out)
    assert got["ok"]
    eng = FlatEncoder(_model(dev), 2, lr=1e-3, weight_decay=0.1, n_buckets=2)
    x = _vols().to(dev)
    p0 = eng.flat.clone()
    for _ in range(2):
        eng.train_step(x)
    torch.cuda.synchronize()
    moved = float((eng.flat - p0).norm())
    diff = float((eng.flat.cpu() - got["flat"]).norm())
    print(f"RCCL world-1 run vs plain run: param movement {moved:.3e}, difference {diff:.3e}")
    # same math; split-K float atomics sum in a different order run to run, and AdamW's first steps (update ~ lr * g / |g|) turn
    # last-bit gradient differences of near-zero gradients into visible ones
    assert moved > 0 and diff < 2e-2 * moved


# ---- DualEngine (configs[2], bench.py's default workload) across ranks: bucket all-reduces launched from BOTH engine streams ----
def _dual_parts(dev, layers=4):
    from oracle import ref_cpu as R
    from test_gpu_engine import _mine, _oracle
    _, _, _, sds, x, s = _oracle(2, 950, layers)
    return _mine(sds, dev, layers), x, s                     # eval(): the Dropout(0.1) sites would draw different masks per rank


def _dual_worker(rank, world, port, out, two_streams, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hsenet_amd.engine import DualEngine
        (tower, p1, p2), x, s = _dual_parts(dev)
        eng = DualEngine(tower, p1, p2, 1, lr=1e-3, weight_decay=0.1, n_buckets=2, two_streams=two_streams)
        assert eng.buckets.world == 2 and len(eng.buckets.bounds) == 2 + 2 + 2
        assert (eng.side is not None) == two_streams
        xr, sr = x[rank:rank + 1].to(dev), s[rank:rank + 1].to(dev)
        for _ in range(2):
            eng.train_step(xr, sr)
        torch.cuda.synchronize()
        torch.save(eng.flat.cpu(), f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


def _dual_single(dev):
    from hsenet_amd.engine import DualEngine
    (tower, p1, p2), x, s = _dual_parts(dev)
    eng = DualEngine(tower, p1, p2, 2, lr=1e-3, weight_decay=0.1, two_streams=False)
    p0 = eng.flat.clone()
    for _ in range(2):
        eng.train_step(x.to(dev), s.to(dev))
    torch.cuda.synchronize()
    return eng.flat.cpu(), float((eng.flat - p0).norm())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("two_streams", [False, True])
def test_dual_engine_two_ranks_match_single_rank(dev, tmp_path, two_streams):
    """2 ranks x 1 volume == 1 rank x 2 volumes for the dual-encoder step, with the 2E3 chain (and ITS bucket all-reduces) on the
    side stream and without; the two replicas must end bit-identical."""
    out = str(tmp_path / f"dual2_{int(two_streams)}")
    mp.spawn(_dual_worker, args=(2, _free_port(), out, two_streams, "gloo"), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0, r1)
    single, moved = _dual_single(dev)
    diff = float((single - r0).norm())
    print(f"dual step (two_streams={two_streams}): param movement {moved:.3e}, |dp2 - single| {diff:.3e}")
    assert moved > 0 and diff < 3e-2 * moved


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("two_streams", [False, True])
def test_dual_engine_two_gpus_rccl(dev, tmp_path, two_streams):
    """the same equivalence with one process per GPU over RCCL (runs only where two GPUs are visible)."""
    out = str(tmp_path / f"dual2n_{int(two_streams)}")
    mp.spawn(_dual_worker, args=(2, _free_port(), out, two_streams, "nccl"), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0, r1)
    single, moved = _dual_single(dev)
    assert moved > 0 and float((single - r0).norm()) < 3e-2 * moved


def _dual_nccl1_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from hsenet_amd.engine import DualEngine
        (tower, p1, p2), x, s = _dual_parts(dev)
        eng = DualEngine(tower, p1, p2, 2, lr=1e-3, weight_decay=0.1, n_buckets=2, two_streams=True)
        eng.buckets.always = True               # identity all-reduces on RCCL, launched from the side stream AND the main stream
        for _ in range(2):
            eng.train_step(x.to(dev), s.to(dev))
        torch.cuda.synchronize()
        torch.save(eng.flat.cpu(), out)
    finally:
        dist.destroy_process_group()


def test_dual_engine_rccl_world1_two_streams(dev, tmp_path):
    """RCCL stream semantics of the two-stream engine on the one GPU there is: every bucket all-reduce is issued (asynchronously)
    from the stream that produced the range — the 2E3 chain's from the side stream — and waited for on the main stream before
    AdamW; the result must equal the run without a process group."""
    out = str(tmp_path / "dual_nccl1.pt")
    mp.spawn(_dual_nccl1_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    single, moved = _dual_single(dev)
    diff = float((single - got).norm())
    print(f"dual step on RCCL (world 1, two streams): param movement {moved:.3e}, difference {diff:.3e}")
    assert moved > 0 and diff < 2e-2 * moved


@pytest.mark.parametrize("self_launch", [True, False])
@pytest.mark.parametrize("workload,extra", [("dual256", ["--batch", "1"]), ("dual256", ["--scaling", "strong", "--global-batch", "2"]),
                                            ("clip_c4b", []), ("clip_s1", ["--grad-wire", "bf16", "--grad-algo", "rs_ag"])])
def test_bench_two_rank_rehearsal(workload, extra, self_launch):
    """bench.py for N > 1, both ways it can be started: `python bench.py --gpus 2` (no WORLD_SIZE: bench.py starts its two ranks itself,
    as the reference's `accelerate launch` does) and exactly as the driver launches it (python -m torch.distributed.run ... bench.py
    --gpus N).  Two ranks on the one GPU of this box over gloo (HS_BENCH_BACKEND=gloo; on a multi-GPU node the same control flow runs
    over RCCL): per-rank collective-free fit probe, the batch vote, barrier-bracketed timing with the max over ranks, ONE JSON line
    from rank 0, whole-job aggregate."""
    if self_launch and workload != "dual256":
        pytest.skip("the self-launch form is covered on the dual256 cases")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--no-cpu-baseline", "--no-roofline"] + extra
    if self_launch:
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        cmd = [sys.executable, os.path.join(root, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py")] + tail
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    strong = "strong" in extra
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == ("strong" if strong else "weak")
    assert d["config"]["parallelism"] == "dp2" and d["config"]["batch_per_gpu"] == 1 and d["config"]["global_batch"] == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]        # whole-job volumes / max-over-ranks time
    assert d["outputs_finite"] is True
    if "--grad-wire" in extra:
        assert d["config"]["grad_exchange"] == "bf16/rs_ag"


@pytest.mark.parametrize("workload", ["dual256", "clip_c4a"])
def test_bench_three_rank_rehearsal(workload):
    """the N-rank control flow at the largest world this one-GPU box admits (its process guard allows six processes on the GPU, the test runner and
    the launcher count, and an odd world exercises the rank arithmetic; the reference runs eight ranks, README.md:78-105 -- world 8 itself is covered on the CPU over gloo,
    tests/test_dp_gloo.py): `python bench.py --gpus 3 --batch 1` starts its three ranks itself, over gloo on the one GPU -- per-rank fit
    probe and vote, ten gradient buckets exchanged in backward-completion order (C2: one fused all-gather of 3 x [1, 1536] features for
    the CLIP step), max-over-ranks timing, ONE line with the whole-job aggregate, identical replicas, and the communication summary
    (`comm`) every N > 1 line carries so that the first real multi-GPU run diagnoses itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--batch", "1", "--steps", "2", "--warmup", "1", "--workload", workload,
           "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["config"]["parallelism"] == "dp3" and d["config"]["batch_per_gpu"] == 1 and d["config"]["global_batch"] == 3
    assert d["outputs_finite"] is True and d["replicas_equal"] is True
    assert d["value"] > 0 and abs(d["value"] - 3 * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]
    c = d["comm"]
    want = {"dual256": 2 * (87389184 + 3.9e6) + 2 * 14166528, "clip_c4a": 89753857 + 3.9e6}[workload] * 4       # fp32 wire
    assert c["wire"] == "fp32" and c["algo"] == "allreduce" and abs(c["comm_bytes_per_step"] - want) < 0.05 * want
    assert c["buckets"] >= 4 and "comm_ms_exposed" in c


@pytest.mark.parametrize("batch", [2, 3])
def test_bench_small_explicit_batches_print_a_line(batch):
    """`--batch 2` / `--batch 3` (the reference's own per-device batches) with the batch legs on: the legs skip every batch >= the one asked
    for and the line is still printed (round 5 raised KeyError after all timed legs had run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", str(batch), "--steps", "2", "--warmup", "1", "--no-extras",
                        "--no-cpu-baseline", "--no-roofline"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["batch_per_gpu"] == batch and "batch1" in d and f"batch{batch}" not in d and ("batch2" in d) == (batch == 3)
    assert all("what" in d[k] for k in d if k.startswith("batch") and k[5:].isdigit())
