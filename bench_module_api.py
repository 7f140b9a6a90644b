#!/usr/bin/env python3
"""bench_module_api.py — the train step a DROP-IN user gets: the reference's own call pattern on hsenet_amd's module API.

The reference trains through ``nn.Module.forward`` -> autograd -> ``torch.optim.AdamW`` under HF Trainer (optim "adamw_torch",
weight_decay 0.1, max_grad_norm 1.0: Preprint/LaMed/src/train/train_CLIP_stage1.py:85,88-92,231-244; train_VLM.py:147-153;
lamed_arch.py:64-76).  `bench.py`'s headline times `engine.DualEngine.train_step` (flat buffers, own AdamW, no autograd graph) — an API
the reference's scripts never call.  This leg times the SAME step definition (BASELINE configs[2]: dual encoders + two Spatial Packers
in train(), L = mean(out^2), clipped AdamW, the same synthetic inputs) written the way the reference writes it:

    out  = hsenet_amd.encode_images(tower, packer1, packer2, vol, slices)      # nn.Module.forward over autograd Functions
    loss = out.square().mean();  loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0);  torch.optim.AdamW(fused=True).step();  zero_grad(set_to_none=True)

and reports it beside the engine's rate on the same box, batch and inputs (`vs_engine`).  fp16: torch.amp.GradScaler around the same
step (the loss scale / unscale / inf check a torch fp16 user runs).

    python bench_module_api.py --batch 1 3 13 --precision bf16 f16      (one JSON object on stdout; also called by bench.py)
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
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OUT_DIM = 3072
GEOM = dict(img=(256, 256, 256), patch=(16, 16, 16), grid=(16, 16, 16), kernel=(2, 4, 4))


class TowerCfg:
    vision_tower = "vit_stage2_dual_encoders"
    vision_select_layer, vision_select_feature, image_channel = -1, "patch", 1
    remain_2d3d_ViT_type = "dual_vits"

    def __init__(self, img, patch):
        self.image_size, self.patch_size = img, patch


def build_modules(geom, dev, seed=1234):
    import hsenet_amd as H
    torch.manual_seed(seed)
    img, patch = geom["img"], geom["patch"]
    tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev)
    pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
    p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
    p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
    return tower, p1, p2


def hf_param_groups(mods, weight_decay):
    """HF Trainer's two groups (get_decay_parameter_names): everything except LayerNorm parameters and biases decays."""
    from hsenet_amd.engine import decays
    dec, nodec = [], []
    for m in mods:
        d = decays(m)
        for n, p in m.named_parameters():
            if p.requires_grad:
                (dec if d[n] else nodec).append(p)
    return [{"params": dec, "weight_decay": weight_decay}, {"params": nodec, "weight_decay": 0.0}]


def synthetic_inputs(geom, batch, dev, rank=0):
    """the same draws as bench.py's inputs(): default_rng(1234 + rank), two sets, one volume at a time."""
    img = geom["img"]
    rng = np.random.default_rng(1234 + rank)
    vols = []
    for _ in range(2):
        v = torch.empty((batch, 1) + img, dtype=torch.float32, device=dev)
        for i in range(batch):
            v[i, 0].copy_(torch.from_numpy(rng.random(img, dtype=np.float32)))
        vols.append(v)
    slices = [torch.from_numpy(rng.standard_normal((batch, 32, 768), dtype=np.float32)).to(dev) for _ in range(2)]
    return vols, slices


def module_step_fn(mods, precision, lr=1e-4, weight_decay=0.1, max_grad_norm=1.0, fused=True):
    """-> step(vol, slices): one train step through the module API, torch autograd and torch's AdamW."""
    import hsenet_amd as H
    tower, p1, p2 = mods
    for m in mods:
        m.train()
    params = [p for m in mods for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(hf_param_groups(mods, weight_decay), lr=lr, betas=(0.9, 0.999), eps=1e-8, fused=fused)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_interval=10 ** 9) if precision == "f16" else None

    def step(vol, s2d):
        out = H.encode_images(tower, p1, p2, vol, s2d)
        loss = out.square().mean()
        if scaler is None:
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
            opt.step()
        else:
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
            scaler.step(opt)
            scaler.update()
        opt.zero_grad(set_to_none=True)
        return loss
    return step, params


def time_steps(step, vols, slices, steps, warmup):
    for i in range(warmup):
        step(vols[i % 2], slices[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(vols[i % 2], slices[i % 2])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def engine_rate(geom, batch, dev, precision, steps, warmup, vols, slices):
    from hsenet_amd.engine import DualEngine
    tower, p1, p2 = build_modules(geom, dev)
    eng = DualEngine(tower, p1, p2, batch, lr=1e-4, weight_decay=0.1, precision=precision)
    dt = time_steps(lambda v, s: eng.train_step(v, s), vols, slices, steps, warmup)
    fin = bool(torch.isfinite(eng.flat).all().item())
    del eng
    return batch * steps / dt, fin


def module_rate(geom, batch, dev, precision, steps, warmup, vols, slices):
    import hsenet_amd as H
    mods = build_modules(geom, dev)
    H.set_precision(precision)
    try:
        step, params = module_step_fn(mods, precision)
        dt = time_steps(step, vols, slices, steps, warmup)
        fin = all(bool(torch.isfinite(p).all().item()) for p in params[:8])
    finally:
        H.set_precision("bf16")
    return batch * steps / dt, fin


def steps_for(batch, steps):
    """timed steps at this per-GPU batch: at least `steps`, and enough of them for ~0.5 s of timed work (a one-volume step is 15 ms:
    five of them are inside the run-to-run noise)."""
    return max(steps, int(round(36.0 / batch)))


def module_api_train(dev, batches=(1, 3, 13), precisions=("bf16", "f16"), steps=6, warmup=3, geom=GEOM, engine_rates=None):
    """the bench leg: {"b<batch>": {"bf16": {module, engine, vs_engine}, "f16": {...}}}.  engine_rates: {(batch, precision): volumes/s}
    already measured by the caller on this box (bench.py's batch legs) are reused instead of being timed again."""
    res = {"unit": "CT-volumes/sec", "steps": {f"b{b}": steps_for(b, steps) for b in batches}, "warmup": warmup,
           "what": "BASELINE configs[2] step through the DROP-IN module API the reference's scripts call: hsenet_amd.encode_images(tower, "
                   "packer, packer2, vol, slices) in train(), loss = out.square().mean(), loss.backward() (torch autograd over the C-ABI "
                   "Functions), torch.nn.utils.clip_grad_norm_(1.0), torch.optim.AdamW(fused=True, HF Trainer's decay / no-decay groups), "
                   "zero_grad(set_to_none=True); f16 = the same under torch.amp.GradScaler; `engine` = engine.DualEngine.train_step on the "
                   "same box / batch / inputs"}
    for b in batches:
        vols, slices = synthetic_inputs(geom, b, dev)
        row = {}
        for prec in precisions:
            torch.cuda.empty_cache()
            mod, fin = module_rate(geom, b, dev, prec, steps_for(b, steps), warmup, vols, slices)
            torch.cuda.empty_cache()
            eng = (engine_rates or {}).get((b, prec))
            if eng is None:
                eng, _ = engine_rate(geom, b, dev, prec, steps_for(b, steps), warmup, vols, slices)
            row[prec] = {"module": round(mod, 3), "engine": round(eng, 3), "vs_engine": round(mod / eng, 4), "finite": fin}
            print(f"[module_api_train] batch {b} {prec}: module {mod:.2f} engine {eng:.2f} volumes/s ({mod / eng:.3f})", file=sys.stderr, flush=True)
        res[f"b{b}"] = row
        del vols, slices
        torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 3, 13])
    ap.add_argument("--precision", nargs="+", default=["bf16", "f16"], choices=["bf16", "f16"])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("bench_module_api.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from hsenet_amd import _lib as L
    L.load()
    L.call("hs_device_check", 0)
    print(json.dumps({"module_api_train": module_api_train(dev, a.batch, a.precision, a.steps, a.warmup)}), flush=True)


if __name__ == "__main__":
    main()
