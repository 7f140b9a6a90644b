#!/usr/bin/env python3
"""bench.py — CT-volumes/sec of the 3D-ViT encoder train step on MI355X (BASELINE.json metric).

Workload at N GPUs (weak scaling, one process per GPU): BASELINE configs[1] — 256^3 volume, 3D-ViT
(patch 16^3, 12 layers, d=768, 12 heads, N=4097 tokens), fwd + bwd + DP gradient all-reduce (RCCL) + AdamW,
bf16 MFMA inputs / fp32 accumulate / fp32 residual stream, `--batch` volumes per GPU, synthetic U[0,1) volumes
already resident in HBM, random-init weights.  A "step" = one such train step over one batch.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel class, measured with HIP events on
the launch stream in a second, instrumented pass over the same steps) and `cpu_baseline` (the CPU oracle —
a plain-PyTorch fp32 port of the reference — timed on this host, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X (MI355X_MICROARCH.md "Chip-level parameters")
PEAK_HBM_GBS = 8000.0

GEOM = {
    "vit256": dict(img=(256, 256, 256), patch=(16, 16, 16)),    # BASELINE configs[1]
    "native": dict(img=(32, 256, 256), patch=(4, 16, 16)),      # the reference's own geometry
}


def flops_per_volume(Np, Pd, d, L):
    """fwd = 2 Np Pd d + L (24 N d^2 + 4 N^2 d); fwd+bwd = 3x."""
    N = Np + 1
    fwd = 2.0 * Np * Pd * d + L * (24.0 * N * d * d + 4.0 * N * N * d)
    return fwd, 3.0 * fwd


def measured_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from committed rocprofv3 --pmc passes (profiles/r01_traffic.json, produced by
    tools/collect_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as the MI355X microarch
    guide prescribes).  None when no measurement for this kernel / batch is committed."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        e = t.get(kernel)
        if e and int(e.get("batch", -1)) == int(batch):
            return e["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(geom, threads, max_seconds=60.0):
    """the oracle (port of the reference's CPU path) on the host cores: 1 volume fwd+bwd, fp32."""
    from oracle import ref_cpu as R
    torch.set_num_threads(threads)
    img, patch = geom["img"], geom["patch"]
    m = R.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True)
    m.load_state_dict(R.seeded_state_dict(m, 41))
    x = R.synthetic_volume((1, 1) + img, 1234)
    t0 = time.perf_counter()
    y, _ = m(x)
    (y * y).mean().backward()
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="CT-volumes/sec", cores=threads, kind="port",
                sample=f"1 volume {img} fwd+bwd, fp32, torch {torch.__version__} CPU oracle (oracle/ref_cpu.py), "
                       f"{threads} threads, single un-warmed iteration = {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=20, help="volumes per GPU per step")
    ap.add_argument("--workload", default="vit256", choices=list(GEOM))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    # HS_BENCH_BACKEND=gloo + fewer visible GPUs than ranks: rehearsal of the N>1 control flow on a 1-GPU box
    backend = os.environ.get("HS_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if a.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import hsenet_amd as H
    from hsenet_amd import _lib as L
    from hsenet_amd.engine import FlatEncoder, profile_classes
    L.load()
    L.call("hs_device_check", local)

    geom = GEOM[a.workload]
    img, patch = geom["img"], geom["patch"]
    torch.manual_seed(1234)
    model = H.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True).to(dev)
    eng = FlatEncoder(model, a.batch, lr=1e-4, weight_decay=0.1)
    rng = np.random.default_rng(1234 + rank)
    vols = [torch.from_numpy(rng.random((a.batch, 1) + img, dtype=np.float32)).to(dev) for _ in range(2)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        eng.train_step(vols[i % 2])
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.train_step(vols[i % 2])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(eng.y).all().item())

    # ---- instrumented pass: per-kernel-class HIP-event timing on the launch stream, same steps --------------
    # EVERY rank runs these steps (they contain the gradient all-reduces); only rank 0 records and reads the timers.
    roof, classes = None, []
    if not a.no_roofline:
        if rank == 0:
            L.call("hs_prof_enable", 1)
        for i in range(max(1, min(a.steps, 3))):
            eng.train_step(vols[i % 2])
        torch.cuda.synchronize()
        if rank == 0:
            classes = profile_classes()
            L.call("hs_prof_enable", 0)
            for c in classes:
                c["avg_us"] = 1e3 * c["total_ms"] / c["launches"]
                c["tflops"] = c["flops"] / (c["total_ms"] * 1e-3) / 1e12 if c["flops"] else 0.0
                c["gbs"] = c["bytes"] / (c["total_ms"] * 1e-3) / 1e9
            mf = [c for c in classes if c["flops"]]
            dom = max(mf, key=lambda c: c["total_ms"])
            roof = dict(kernel=dom["name"], bound="mfma", achieved=round(dom["tflops"], 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                        frac=round(dom["tflops"] / PEAK_BF16_TFLOPS, 4), traffic=measured_traffic(dom["name"], a.batch),
                        avg_launch_us=round(dom["avg_us"], 2), flops_per_launch=dom["flops"] / dom["launches"])
    if world > 1:
        dist.barrier()

    Np = model.patch_embedding.n_patches
    fwd, fb = flops_per_volume(Np, model.patch_embedding.patch_dim, model.hidden_size, len(model.blocks))
    vols_per_s = world * a.batch * a.steps / dt
    out = {
        "metric": "CT-volumes/sec (256\u00b3) dual-encoder fwd+bwd @1/2/4/8 MI355X; % MFMA roofline", "value": round(vols_per_s, 3), "unit": "CT-volumes/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {img[0]}x{img[1]}x{img[2]} volume, 3D-ViT patch {patch} 12 layers d=768 "
                               f"(N={Np + 1} tokens), fwd+bwd+grad all-reduce+AdamW, random-init weights",
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}"},
        "model_tflops": round(vols_per_s * fb / 1e12, 2),
        "mfma_roofline_frac_end_to_end": round(vols_per_s * fb / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "outputs_finite": finite,
    }
    if roof:
        out["roofline"] = roof
        out["kernels"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items() if k != "bytes" and k != "flops"}
                          for c in sorted(classes, key=lambda c: -c["total_ms"])]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(geom, threads=min(os.cpu_count() or 1, 16))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
