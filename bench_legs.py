#!/usr/bin/env python3
"""bench_legs.py -- the side legs of bench.py's default line (N = 1 only), each a function of (workload, geometry, device): forward-only
rates per operand format, one-sample latency, the exact-f32 parity mode's rate, the 16-bit modes' distance from it on one train step, the
PCIe-inclusive rate, and the CPU baseline (the oracle -- a plain-PyTorch fp32 port of the reference -- timed on this host's cores).
bench.py owns the headline, the roofline pass, the batch legs and the JSON line; bench_module_api.py the drop-in train step."""
from __future__ import annotations

import os
import platform
import sys
import time

import numpy as np
import torch

from bench_module_api import OUT_DIM, TowerCfg


def eval_forward_rates(workload, geom, dev, batch=4, steps=3):
    """forward-only volumes/s of the module API (torch.no_grad(), eval(): the reference's generation / eval path runs the encoders once per
    sample) at each operand precision: bf16 (the training dtype), fp16 (the reference's eval-time autocast dtype,
    Bench/eval/eval_HSENet_CT_Rate_MRG.py:383,423: same kernels with fp16 packs, outputs within 1e-3 of the reference) and exact f32."""
    import hsenet_amd as H
    img, patch = geom["img"], geom["patch"]
    torch.manual_seed(1234)
    rng = np.random.default_rng(98)
    vol = torch.from_numpy(rng.random((batch, 1) + img, dtype=np.float32)).to(dev)
    s2d = torch.from_numpy(rng.standard_normal((batch, 32, 768), dtype=np.float32)).to(dev)
    if workload in ("dual256", "native_dual"):
        tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev).eval()
        pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
        p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev).eval()
        p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev).eval()
        fwd = lambda: H.encode_images(tower, p1, p2, vol, s2d)
    elif workload == "vit256":
        m = H.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True).to(dev).eval()
        fwd = lambda: m(vol)[0]
    else:
        return None
    res = {"batch_per_gpu": batch, "steps": steps, "unit": "CT-volumes/sec",
           "what": "forward only, module API under torch.no_grad() (the eval / generation path: encoders once per sample)"}
    ref = None
    for prec in ("f32", "bf16", "f16"):
        H.set_precision(prec)
        try:
            with torch.no_grad():
                y = fwd()                                       # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    y = fwd()
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        finally:
            H.set_precision("bf16")
        res[prec] = round(batch / dt, 3)
        if prec == "f32":
            ref = y.double()
        else:
            res[prec + "_rel_err_vs_f32"] = float(f"{float((y.double() - ref).norm() / ref.norm()):.3e}")
    return res


def eval_latency(workload, geom, dev, reps=10):
    """ms of ONE sample through encoders + packers at fp16 operands under torch.no_grad(): the generation path runs the vision side once
    per sample (language_model/lamed_phi3.py:151-228) under torch.cuda.amp.autocast() = fp16 (Bench/eval/eval_HSENet_CT_Rate_MRG.py:383,423)."""
    import hsenet_amd as H
    if workload not in ("dual256", "native_dual"):
        return None
    img, patch = geom["img"], geom["patch"]
    torch.manual_seed(1234)
    rng = np.random.default_rng(96)
    vol = torch.from_numpy(rng.random((1, 1) + img, dtype=np.float32)).to(dev)
    s2d = torch.from_numpy(rng.standard_normal((1, 32, 768), dtype=np.float32)).to(dev)
    tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev).eval()
    pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
    p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev).eval()
    p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev).eval()
    res = {}
    for prec in ("f16", "bf16"):
        H.set_precision(prec)
        try:
            with torch.no_grad():
                for _ in range(2):
                    H.encode_images(tower, p1, p2, vol, s2d)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    H.encode_images(tower, p1, p2, vol, s2d)
                torch.cuda.synchronize()
            res[prec] = round(1e3 * (time.perf_counter() - t0) / reps, 3)
        finally:
            H.set_precision("bf16")
    return {"f16": res["f16"], "bf16": res["bf16"], "unit": "ms per sample", "batch": 1, "reps": reps,
            "what": "one volume -> dual encoders -> two Spatial Packers -> (1,256,3072), module API under torch.no_grad(): the vision side of "
                    "one generate() call"}


def f32_mode_rate(workload, geom, dev, steps=2):
    """volumes/s of the exact-f32 parity mode (hsenet_amd.set_precision("f32"): v_mfma_f32_32x32x2_f32 GEMMs and attention, the mode that
    meets the north-star's <= 1e-3) for the same modules, forward + backward of one volume through the module API (autograd)."""
    import hsenet_amd as H
    img, patch = geom["img"], geom["patch"]
    torch.manual_seed(1234)
    rng = np.random.default_rng(99)
    vol = torch.from_numpy(rng.random((1, 1) + img, dtype=np.float32)).to(dev)
    s2d = torch.from_numpy(rng.standard_normal((1, 32, 768), dtype=np.float32)).to(dev)
    if workload in ("dual256", "native_dual"):
        tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev)
        pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
        p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
        p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
        mods = [tower, p1, p2]

        def step():
            out = H.encode_images(tower, p1, p2, vol, s2d)
            (out * out).mean().backward()
    elif workload == "vit256":
        m = H.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True).to(dev)
        mods = [m]

        def step():
            y, _ = m(vol)
            (y * y).mean().backward()
    else:
        return None
    H.set_precision("f32")
    try:
        step()                                              # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for m in mods:
                m.zero_grad(set_to_none=True)
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        H.set_precision("bf16")
    return {"value": round(1.0 / dt, 3), "unit": "CT-volumes/sec", "ms_per_step": round(1e3 * dt, 2), "batch_per_gpu": 1, "steps": steps,
            "what": "forward + backward, module API (autograd), exact-f32 MFMA everywhere (the mode whose outputs are within 1e-3 of the "
                    "reference: tests/test_gpu_parity.py); no optimizer step"}


def train_accuracy_vs_f32(workload, geom, dev):
    """forward output and one block's qkv weight gradient of ONE volume's train step (L = mean(out^2)) through the module API at each
    16-bit operand format, as distances from the exact-f32 mode on the same volume and weights (that mode is within 1e-5 of the
    reference's fp32 CPU path at full size: tests/test_gpu_grad_goldens.py).  fp16: loss scaled by 2^16 and the gradient unscaled."""
    import hsenet_amd as H
    img, patch = geom["img"], geom["patch"]
    torch.manual_seed(1234)
    rng = np.random.default_rng(97)
    vol = torch.from_numpy(rng.random((1, 1) + img, dtype=np.float32)).to(dev)
    s2d = torch.from_numpy(rng.standard_normal((1, 32, 768), dtype=np.float32)).to(dev)
    if workload in ("dual256", "native_dual"):
        tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev)
        pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
        p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
        p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
        mods = [tower.eval(), p1.eval(), p2.eval()]
        fwd = lambda: H.encode_images(tower, p1, p2, vol, s2d)
        probes = {"3d_vit.blocks.0.qkv": tower.vision_tower_stage1.blocks[0].attn.qkv.weight,
                  "2e3_vit.blocks.11.fc1": tower.vision_tower_stage2.blocks[-1].mlp.linear1.weight, "packer.Wk": p1.resolution_attention.Wk.weight}
    elif workload == "vit256":
        m = H.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True).to(dev).eval()
        mods = [m]
        fwd = lambda: m(vol)[0]
        probes = {"blocks.0.qkv": m.blocks[0].attn.qkv.weight, "blocks.11.fc1": m.blocks[-1].mlp.linear1.weight}
    else:
        return None
    got = {}
    for prec, ls in (("f32", 1.0), ("bf16", 1.0), ("f16", 65536.0)):
        H.set_precision(prec)
        try:
            for m_ in mods:
                m_.zero_grad(set_to_none=True)
            y = fwd()
            ((y * y).mean() * ls).backward()
            torch.cuda.synchronize()
        finally:
            H.set_precision("bf16")
        got[prec] = (y.detach().double(), {k: (p.grad.detach().double() / ls) for k, p in probes.items()})
    rel = lambda a, b: float((a - b).norm() / b.norm())
    res = {"what": "one volume, module API, L = mean(out^2): distance of each 16-bit mode from the exact-f32 mode (forward output; worst of "
                   f"the probed weight gradients {sorted(probes)})"}
    for prec in ("bf16", "f16"):
        res[prec + "_fwd_rel_err"] = float(f"{rel(got[prec][0], got['f32'][0]):.3e}")
        res[prec + "_grad_rel_err"] = float(f"{max(rel(got[prec][1][k], got['f32'][1][k]) for k in probes):.3e}")
    return res


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor() or "unknown CPU"


def host_cores():
    """cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each lease a share
    of a much larger host; oversubscribing torch's thread pool beyond that share makes the CPU leg many times slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(workload, geom, cap_seconds=150.0):
    """the oracle (port of the reference's CPU path) on ALL host cores for the SAME workload: one volume fwd+bwd in fp32,
    1 warm-up + up to 3 timed iterations (fewer only if the warm-up shows they would exceed `cap_seconds`), median.  "All cores" =
    host_cores(): the cores the lease may use, stated in the result."""
    from oracle import ref_cpu as R
    threads = host_cores()
    torch.set_num_threads(threads)
    print(f"[bench] cpu_baseline: {threads} usable cores of {os.cpu_count()} visible", file=sys.stderr, flush=True)
    img, patch = geom["img"], geom["patch"]
    x = R.synthetic_volume((1, 1) + img, 1234)
    if workload in ("dual256", "native_dual"):
        tower = R.ViT3DTower_dual_encoders(TowerCfg(img, patch))
        pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
        p1 = R.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk)
        p2 = R.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk)
        mods = [tower, p1, p2]
        s = R.synthetic_slices(1, 1235)

        def step():
            out = R.encode_images(tower, p1, p2, x, s)
            (out * out).mean().backward()
        what = "dual encoders + 2 Spatial Packers"
    elif workload == "clip_c4a":
        from oracle import clip_cpu as RC
        step, mods = RC.c4a_cpu_step(img, patch, x)
        what = "stage-2 CLIP step (frozen 3D-ViT fwd + 2E3 fwd+bwd + InfoNCE + relation MSE)"
    elif workload == "clip_c4b":
        from oracle import clip_cpu as RC
        step, mods = RC.c4b_cpu_step(img, patch, x)
        what = "both-trainable CLIP step (3D-ViT fwd+bwd + InfoNCE; 2E3 fwd+bwd + InfoNCE + relation MSE)"
    elif workload == "clip_s1":
        from oracle import clip_cpu as RC
        m = RC.M3DCLIPVision_stage1(img, patch)
        t1 = RC.synthetic_text_features(x.shape[0], 1236)
        mods = [m]

        def step():
            m(x, t1)["loss"].backward()
        what = "stage-1 CLIP step (3D-ViT fwd+bwd + InfoNCE)"
    else:
        m = R.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True)
        mods = [m]

        def step():
            y, _ = m(x)
            (y * y).mean().backward()
        what = "3D-ViT"
    for i, m in enumerate(mods):
        m.load_state_dict(R.seeded_state_dict(m, 41 + i))
    t0 = time.perf_counter()
    step()                                           # warm-up (allocator, thread pool, first-touch)
    warm = time.perf_counter() - t0
    print(f"[bench] cpu_baseline: warm-up iteration {warm:.1f} s", file=sys.stderr, flush=True)
    # bounded sample: one volume is ~25 s of CPU work on 16 cores -- ONE timed iteration after the warm-up then (a median of 3 only when
    # an iteration is short), so that the default run stays inside a few minutes
    n = 1 if warm > 10.0 else max(1, min(3, int((cap_seconds - warm) // max(warm, 1e-3))))
    times = []
    for _ in range(n):
        for m in mods:
            m.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline: timed iteration {times[-1]:.1f} s", file=sys.stderr, flush=True)
    med = float(np.median(times))
    return dict(value=1.0 / med, unit="CT-volumes/sec", cores=threads, kind="port",
                sample=f"1 volume {img} fwd+bwd of {what}, fp32, torch {torch.__version__} CPU oracle (oracle/ref_cpu.py) on "
                       f"{cpu_model()} with {threads} threads: 1 warm-up ({warm:.1f} s) + {n} timed, median {med:.1f} s")


def pcie_inclusive_rate(eng, needs_2d, vols, slices, dev, sync, steps):
    """volumes/s when every step's volumes come from pinned host memory (the boundary a data loader hands over: fp32 (B,1,D,H,W) host
    tensors, dataset/multi_dataset.py): double-buffered hipMemcpyAsync on a copy stream, the upload of step i + 1 beside the compute of
    step i.  Also the bare host-to-device rate of one such upload."""
    B = vols[0].shape[0]
    host = [torch.empty(vols[0].shape, dtype=torch.float32).pin_memory() for _ in range(2)]
    for h, v in zip(host, vols):
        h.copy_(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vols[0].copy_(host[0], non_blocking=True)
    torch.cuda.synchronize()
    h2d_gbs = host[0].numel() * 4 / (time.perf_counter() - t0) / 1e9
    from hsenet_amd.fused import _pick_concurrent_stream
    copy = _pick_concurrent_stream(dev)            # (a stream on another hardware queue than the compute stream's: see fused.py)
    main = torch.cuda.current_stream()
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    freed = [torch.cuda.Event(), torch.cuda.Event()]
    for k in range(2):
        freed[k].record(main)

    def upload(k):
        with torch.cuda.stream(copy):
            copy.wait_event(freed[k])                  # the step that read buffer k has been enqueued and has consumed it
            vols[k].copy_(host[k], non_blocking=True)
            ready[k].record(copy)

    def step(i):
        k = i % 2
        main.wait_event(ready[k])
        if needs_2d:
            eng.train_step(vols[k], slices[k])
        else:
            eng.train_step(vols[k])
        freed[k].record(main)
        upload(k)                                      # refill buffer k for step i + 2 while step i + 1 computes
    upload(0), upload(1)
    step(0), step(1)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    main.wait_event(ready[0]), main.wait_event(ready[1])
    torch.cuda.synchronize()
    return {"value": round(B * steps / dt, 3), "unit": "CT-volumes/sec", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
            "h2d_gbs": round(h2d_gbs, 1), "upload_mb_per_step": round(host[0].numel() * 4 / 1e6, 1),
            "what": "the same step with every batch uploaded from pinned host memory (fp32 volumes, double-buffered copy stream beside the "
                    "compute); `value` of the line is the HBM-resident number"}


