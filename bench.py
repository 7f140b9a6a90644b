#!/usr/bin/env python3
"""bench.py — CT-volumes/sec of the dual 3D vision-encoder train step on MI355X (BASELINE.json metric).

Default workload at N GPUs (weak scaling, one process per GPU) = BASELINE configs[2]: 256^3 volume -> 3D-ViT AND 2E3-ViT
(patch 16^3, 12 layers, d=768, 12 heads, N=4097 tokens each; the 2E3 encoder with its slice-guided scorer over (32,768) slice
features) -> two Spatial Packers (grid 16^3, kernel (2,4,4)) -> (B, 256, 3072); loss = mean-square of that output;
fwd + bwd of all four modules + DP gradient all-reduce (RCCL) + clipped AdamW; bf16 MFMA inputs / fp32 accumulate / fp32
residual stream; `--batch` volumes per GPU (default 53: see DEFAULT_BATCH), synthetic U[0,1) volumes and N(0,1) slice features already resident in HBM,
random-init weights.  A "step" = one such train step over one batch.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts the N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Other workloads: --workload vit256 (configs[1]: the 3D-ViT alone), clip_c4a (configs[3]: the reference's stage-2 CLIP step),
mrg_c5 (configs[4]: frozen towers + trainable packers spliced into a random-init HF Phi-4-mini-shaped decoder; reported, not
roofline-scored: the LLM is outside the HIP path), native_dual (configs[2] at the reference's own 32x256x256 / 4x16x16 geometry).

Flags: --scaling strong (fixed GLOBAL batch --global-batch, per-GPU batch = global / world; default weak = fixed per-GPU batch);
--attn-logit-std S (scale the q rows of every qkv weight so that the scaled attention logits have standard deviation S: random-init
attention is nearly uniform, trained attention is peaky and the attention kernels are data-dependent at the power cap).

Prints ONE JSON line (rank 0).  Beside the contract keys: `attention` (credited TFLOP/s and fraction of peak over attention forward +
both backward sweeps: the north-star's own number), `optimizer_excluded` (the rate of forward + backward + all-reduce alone),
`peaky_attention` (the same step and attention figures at logit std 1.5), `f32_mode` (volumes/s of the <= 1e-3 parity mode through
the module API), `eval_forward` (forward-only volumes/s at bf16 / fp16 / f32 operands with the fp16 and bf16 outputs' distance from
the f32 ones: fp16 is the reference's eval-time autocast dtype and lands inside 1e-3 at the bf16 kernels' speed).  Extra objects: `roofline` (dominant kernel class, measured with HIP events on the launch
stream in a second, instrumented pass over the same steps), `cpu_baseline` (the CPU oracle — a plain-PyTorch fp32 port of the
reference — timed on this host for the SAME workload, rank 0, N=1 only) and `batch1` (the same step at 1 volume per GPU, the
per-GPU load of configs[3]/[4]).
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench_module_api import OUT_DIM, TowerCfg  # noqa: E402,F401
from bench_legs import (cpu_baseline, eval_forward_rates, eval_latency, f32_mode_rate, pcie_inclusive_rate,  # noqa: E402,F401
                        train_accuracy_vs_f32)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X (MI355X_MICROARCH.md "Chip-level parameters")
PEAK_HBM_GBS = 8000.0
METRIC = "CT-volumes/sec (256³) dual-encoder fwd+bwd @1/2/4/8 MI355X; % MFMA roofline"

GEOM = {
    "dual256": dict(img=(256, 256, 256), patch=(16, 16, 16), grid=(16, 16, 16), kernel=(2, 4, 4), cfg="configs[2]"),
    "vit256": dict(img=(256, 256, 256), patch=(16, 16, 16), cfg="configs[1]"),
    "clip_c4a": dict(img=(256, 256, 256), patch=(16, 16, 16), cfg="configs[3]"),
    "clip_c4b": dict(img=(256, 256, 256), patch=(16, 16, 16), cfg="configs[3], both encoders trainable (SURVEY 8(d) C4-B)"),
    "clip_s1": dict(img=(256, 256, 256), patch=(16, 16, 16), cfg="the stage-1 CLIP step (CLIP_stage1.py:111-139) at configs[3]'s geometry"),
    "native_dual": dict(img=(32, 256, 256), patch=(4, 16, 16), grid=(8, 16, 16), kernel=(1, 4, 4), cfg="configs[2] at the reference-native geometry"),
    "mrg_c5": dict(img=(256, 256, 256), patch=(16, 16, 16), grid=(16, 16, 16), kernel=(2, 4, 4), cfg="configs[4]"),
}
# Volumes per GPU of the default run.  53 x 33 query blocks x 12 heads = 20 988 attention blocks = 40.99 rounds of the chip's 512 block slots
# (20 volumes: 15.47 rounds -- half a round of tail in every attention launch), 849 row tiles for the GEMMs, ~190 GB of the 288 GB of HBM
# (activations for the backward: 1.3 GB per volume and encoder).  Same-box sweep (round 3): 20 -> 91.1, 31 -> 93.4, 47 -> 94.6, 53 -> 95.4,
# 62 -> 95.5 volumes/s, with the dominant kernel's own rate unchanged or slightly up (dK/dV 870 -> 890 TFLOP/s).
DEFAULT_BATCH = 53
SEQ_LEN = 800     # script/train_vlm_mrg.sh:11 --model_max_length 800


def flops_per_volume(workload, Np, Pd, d=768, L=12, ns=32, cells=128, out=OUT_DIM):
    """encoder fwd = 2 Np Pd d + L (24 N d^2 + 4 N^2 d); scorer (SGA) and packer terms as listed there;
    fwd+bwd = 3 x fwd for everything that is differentiated."""
    N = Np + 1
    enc = 2.0 * Np * Pd * d + L * (24.0 * N * d * d + 4.0 * N * N * d)
    sga = 2 * 2.0 * Np * d * d + 2 * 2.0 * ns * d * d + 4.0 * Np * ns * d
    packer = 2 * 2.0 * Np * d * d + 2 * 2.0 * cells * d * d + 4.0 * Np * d + 2.0 * cells * d * out + 2.0 * cells * out * out
    if workload in ("dual256", "native_dual"):
        return 3.0 * (2 * enc + sga + 2 * packer)
    if workload == "clip_c4a":      # frozen 3D-ViT forward + trainable 2E3 fwd+bwd (CLIP_stage2.py:119-151): 4 forward units
        return enc + 3.0 * (enc + sga)
    if workload == "clip_c4b":      # both trainable: stage-1 step + stage-2 step sharing the stage-1 forward: 6 forward units
        return 3.0 * enc + 3.0 * (enc + sga)
    if workload == "mrg_c5":        # vision side only: frozen towers forward, packers forward + backward (the LLM is not counted)
        return 2 * enc + sga + 3.0 * 2 * packer
    return 3.0 * enc


def measured_pmc(kernel, batch, workload, build_id=None):
    """rocprofv3 --pmc figures of `kernel` at this batch / workload from the newest committed summary (profiles/r0N_pmc.json from
    tools/collect_traffic.sh + tools/collect_util.sh + tools/util_to_json.py): HBM bytes per launch (FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH_SIZE doubled as the MI355X microarch guide prescribes) and the share of matrix-pipe cycles with an MFMA
    executing.  The counters cannot be collected inside the timed run (counter collection serialises kernels), so they are tied to
    the library instead: with `build_id` given, a summary whose `_build_id` differs from the loaded library's is NOT used
    (-> {"stale": file}).  {} when nothing is committed for this kernel / batch / workload."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json"):
        try:
            allk = json.load(open(os.path.join(ROOT, "profiles", name)))
            e = allk.get(kernel)
            if e and int(e.get("batch", -1)) == int(batch) and e.get("workload") == workload and "hbm_bytes_per_launch" in e:
                if build_id is not None and allk.get("_build_id") != build_id:
                    return {"stale": f"profiles/{name}", "build_id": allk.get("_build_id")}
                return dict(e, source=f"profiles/{name}", build_id=allk.get("_build_id"))
        except Exception:
            pass
    return {}


def attention_aggregate(classes, prefix="attn_"):
    """credited TFLOP/s over attention forward + backward (forward 4 N^2 d, backward 8 N^2 d per (batch, head, layer));
    prefix "attn_fwd": the forward alone (the optimistic kernel + its fix-up pass) -- the north-star's ">= 40 % on the 3D-ViT attention"
    holds for the forward and not for forward + backward, and the line says so with both figures."""
    classes = [c for c in classes if c["name"].startswith(prefix)]
    att = [c for c in classes if c["name"].startswith("attn_") and c["flops"]]
    if not att:
        return None
    fl, ms = sum(c["flops"] for c in att), sum(c["total_ms"] for c in classes if c["name"].startswith("attn_"))
    tf = fl / (ms * 1e-3) / 1e12
    return {"credited_tflops": round(tf, 2), "frac": round(tf / PEAK_BF16_TFLOPS, 4), "ms_per_step_share": None,
            "kernels": sorted(c["name"] for c in classes if c["name"].startswith("attn_")), "total_ms": round(ms, 3)}


def encoders_of(eng):
    """the ViT modules an engine trains (for --attn-logit-std)."""
    ms = []
    for nm in ("u1", "u2", "unit", "u"):
        u = getattr(eng, nm, None)
        if u is not None and hasattr(u, "model"):
            ms.append(u.model)
    return ms


def set_attn_logit_std(models, std):
    """scale the q rows of every block's qkv weight so that the scaled logits q.k / 8 have standard deviation `std` for unit-variance
    LayerNorm outputs: std(logit) = sigma_q * sigma_k with sigma^2 = mean squared row norm of Wq / Wk (64-dim heads, 1/8 scaling).
    Returns the factor applied to the first block (random init: sigma_q sigma_k = 0.31, i.e. near-uniform attention)."""
    first = None
    with torch.no_grad():
        for m in models:
            for blk in m.blocks:
                w = blk.attn.qkv.weight
                d = w.shape[1]
                sq = float(w[:d].float().pow(2).sum(1).mean().sqrt())
                sk = float(w[d:2 * d].float().pow(2).sum(1).mean().sqrt())
                f = std / max(sq * sk, 1e-12)
                w[:d].mul_(f)
                first = f if first is None else first
    return first


def build_engine(workload, geom, batch, dev, group=None, precision="bf16"):
    import hsenet_amd as H
    from hsenet_amd.engine import DualEngine, FlatEncoder
    img, patch = geom["img"], geom["patch"]
    # gradient exchange (hsenet_amd/dp.py): wire fp32 | bf16, algorithm allreduce | rs_ag | auto; default = the reference's DDP setting
    dp = dict(grad_wire=os.environ.get("HS_GRAD_WIRE", "fp32"), grad_algo=os.environ.get("HS_GRAD_ALGO", "allreduce"))
    torch.manual_seed(1234)
    if workload in ("dual256", "native_dual"):
        tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev)
        pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
        p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
        p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
        ts = os.environ.get("HS_DUAL_STREAMS")                     # (env: A/B runs only; default = the engine's batch rule)
        eng = DualEngine(tower, p1, p2, batch, lr=1e-4, weight_decay=0.1, group=group, two_streams=None if ts is None else ts != "0",
                         recompute_mlp=bool(os.environ.get("HS_RECOMPUTE_MLP")), precision=precision, **dp)
        pe = tower.vision_tower_stage1.patch_embedding
        return eng, pe, True
    if workload == "clip_s1":
        from hsenet_amd.clip import M3DCLIPVision_stage1
        from hsenet_amd.clip_step import ClipStage1Engine
        model = M3DCLIPVision_stage1(img, patch).to(dev)
        eng = ClipStage1Engine(model, batch, lr=1e-4, weight_decay=0.1, group=group, precision=precision, **dp)
        rng = np.random.default_rng(4321 + int(os.environ.get("RANK", "0")))
        eng.set_text_features(torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((batch, 768), dtype=np.float32)), dim=-1).to(dev))
        return eng, model.vision_encoder.patch_embedding, False
    if workload in ("clip_c4a", "clip_c4b"):
        from hsenet_amd.clip_step import ClipStage2Engine, make_clip_stage2
        model = make_clip_stage2(img, patch).to(dev)
        if workload == "clip_c4b":
            model.stage1_pretrained_CLIP.requires_grad_(True)
        eng = ClipStage2Engine(model, batch, lr=1e-4, weight_decay=0.1, group=group, train_stage1=workload == "clip_c4b", precision=precision, **dp)
        rng = np.random.default_rng(4321 + int(os.environ.get("RANK", "0")))
        txt = [torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((batch, 768), dtype=np.float32)), dim=-1).to(dev)
               for _ in range(2)]                                   # frozen text embeddings: normalize(N(0,1))
        eng.set_text_features(*txt)
        return eng, model.vision_encoder.patch_embedding, True
    if workload == "mrg_c5":
        return build_mrg(geom, batch, dev, group, precision)
    model = H.ViT_stage1(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True).to(dev)
    eng = FlatEncoder(model, batch, lr=1e-4, weight_decay=0.1, group=group, precision=precision, **dp)
    return eng, model.patch_embedding, False


def build_mrg(geom, batch, dev, group, precision="bf16"):
    """BASELINE configs[4]: dual encoders (frozen) + two packers (trainable) spliced into a stock HF Phi-4-mini-shaped decoder
    (random-init from its config -- no weights offline --, bf16, frozen; the LLM runs on PyTorch-ROCm and is OUT OF SCOPE of the HIP
    path: it is there so that the step has the reference's shape, lamed_phi3.py:41-148)."""
    import hsenet_amd as H
    from hsenet_amd.mrg_step import MRGStepEngine
    from transformers import Phi3Config, Phi3ForCausalLM
    img, patch = geom["img"], geom["patch"]
    tower = H.build_vision_tower(TowerCfg(img, patch)).to(dev)
    pk = dict(grid=geom["grid"], kernel_size=geom["kernel"])
    p1 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
    p2 = H.VisualPacker_3d_phi_v3(img, patch, 768, OUT_DIM, "mlp", 2, **pk).to(dev)
    eng = MRGStepEngine(tower, p1, p2, batch, lr=1e-4, weight_decay=0.0, group=group, precision=precision)       # script/train_vlm_mrg.sh:14,26: lr 1e-4, wd 0
    cfg = Phi3Config(hidden_size=OUT_DIM, intermediate_size=8192, num_hidden_layers=32, num_attention_heads=24, num_key_value_heads=8,
                     vocab_size=200064, max_position_embeddings=4096, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                     tie_word_embeddings=True, attn_implementation="sdpa")
    torch.manual_seed(99)
    with torch.device(dev):
        llm = Phi3ForCausalLM(cfg).to(torch.bfloat16)
    llm.requires_grad_(False)
    rng = np.random.default_rng(77 + int(os.environ.get("RANK", "0")))
    ids = torch.from_numpy(rng.integers(3, 200064, (batch, SEQ_LEN))).to(dev)
    labels = ids.clone()
    labels[:, :1 + 256] = -100                                    # no LM loss on the bos + image-token positions
    with torch.no_grad():
        embeds = llm.get_input_embeddings()(ids)

    def llm_loss(emb):
        return llm(inputs_embeds=emb.to(torch.bfloat16), labels=labels).loss

    class _Step:        # adapter to the bench loop's train_step(vol, slices)
        flat = eng.flat

        def train_step(self, vol, s2d):
            eng.train_step(vol, s2d, embeds, llm_loss)
    return _Step(), tower.vision_tower_stage1.patch_embedding, True


def timed_run(eng, needs_2d, vols, slices, steps, warmup, sync, optimizer=True):
    kw = {} if optimizer else {"optimizer": False}

    def step(i):
        if needs_2d:
            eng.train_step(vols[i % 2], slices[i % 2], **kw)
        else:
            eng.train_step(vols[i % 2], **kw)
    for i in range(warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    return time.perf_counter() - t0, step


def visible_gpus():
    """GPUs this process could use, counted WITHOUT initialising the GPU runtime (the N ranks are started from this process: it must not
    hold a GPU context): KFD topology nodes with SIMDs, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES;
    torch.cuda.device_count() (which does not initialise the GPU on this image's amdsmi path) only when the topology is unreadable."""
    n = None
    try:
        root = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(root):
            for line in open(os.path.join(root, node, "properties")):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except Exception:
        n = None
    if not n:
        return torch.cuda.device_count()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks here, as the reference's launcher does
    (script/train_clip_stage1.sh:4 `accelerate launch`, README.md:78-105: 8 processes), and exit with their status.  Called BEFORE anything
    in this process touches the GPU; the ranks are fresh children of `python -m torch.distributed.run --standalone` (which picks its own
    free rendezvous port: no bind-then-close race between concurrent benches), never a re-exec of this process.  Refuses (exit 2) when
    fewer than N devices are visible -- unless HS_BENCH_BACKEND=gloo asks for the one-GPU rehearsal of the N-rank control flow."""
    import subprocess
    have = visible_gpus()
    if have < n and os.environ.get("HS_BENCH_BACKEND", "nccl") == "nccl":
        sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; not printing a line for a different world size\n")
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + list(argv)
    print(f"[bench] starting {n} ranks: {' '.join(cmd[1:8])} bench.py ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help=f"volumes per GPU per step (default: {DEFAULT_BATCH}; clip_c4a / mrg_c5: 1 = global batch 8 on 8 GPUs)")
    ap.add_argument("--workload", default="dual256", choices=list(GEOM))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--recompute-mlp", action="store_true", help="dual256 / native_dual: do not keep the MLP activations (46 %% of the saved "
                    "activation bytes), re-run fc1 + GELU per layer in the backward; config.recompute says so (NOT the default headline)")
    ap.add_argument("--no-batch1", action="store_true")
    ap.add_argument("--no-module-api", action="store_true", help="skip the module_api_train leg (the drop-in train step beside the engine's)")
    ap.add_argument("--no-extras", action="store_true", help="skip the optimizer_excluded / peaky_attention / f32_mode / eval_forward legs")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch volumes on EVERY GPU; strong: --global-batch volumes in total, split over the GPUs")
    ap.add_argument("--global-batch", type=int, default=None, help="--scaling strong: total volumes per step (default 56 = 7 per GPU at 8 GPUs; "
                    "clip_c4a / clip_c4b / mrg_c5: 8, BASELINE configs[3]/[4])")
    ap.add_argument("--grad-wire", default=None, choices=["fp32", "bf16"], help="gradient exchange wire format (default fp32; also HS_GRAD_WIRE)")
    ap.add_argument("--grad-algo", default=None, choices=["allreduce", "rs_ag", "auto"], help="gradient exchange algorithm (default allreduce; also HS_GRAD_ALGO)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16"], help="16-bit operand format of the whole step: bf16 (the reference's "
                    "autocast policy, the headline) or f16 (IEEE fp16 operands + power-of-two loss scale: same kernels and rate, outputs within "
                    "1e-3 of the fp32 reference); the default run reports the f16 step beside the headline as `f16_train`")
    ap.add_argument("--attn-logit-std", type=float, default=None, help="scale the q rows of the qkv weights so that the scaled attention "
                    "logits have this standard deviation (random init: 0.31)")
    a = ap.parse_args()
    if a.gpus is None:
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: the line's n_gpus must be the --gpus that was asked for")
    if a.recompute_mlp:
        os.environ["HS_RECOMPUTE_MLP"] = "1"
    if a.grad_wire:
        os.environ["HS_GRAD_WIRE"] = a.grad_wire
    if a.grad_algo:
        os.environ["HS_GRAD_ALGO"] = a.grad_algo
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    small = a.workload in ("clip_c4a", "clip_c4b", "clip_s1", "mrg_c5")
    if a.scaling == "strong":
        if a.batch is not None:
            sys.exit("--scaling strong takes --global-batch, not --batch")
        g = a.global_batch if a.global_batch is not None else (8 if small else 56)
        if g % world_env:
            sys.exit(f"--global-batch {g} is not divisible by the {world_env} ranks")
        a.batch = g // world_env
    batch_defaulted = a.batch is None
    if a.batch is None:
        a.batch = 1 if small else DEFAULT_BATCH

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

    from hsenet_amd import _lib as L
    from hsenet_amd.engine import profile_classes
    L.load()
    L.call("hs_device_check", local)

    geom = GEOM[a.workload]
    img, patch = geom["img"], geom["patch"]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def inputs(batch):
        """two sets of `batch` synthetic volumes U[0,1) + slice features N(0,1), resident in HBM (default_rng(1234 + rank)).
        Drawn and uploaded one volume (64 MiB) at a time: the same values as one (batch, 1, D, H, W) draw, without a multi-GB host array per
        rank (53 volumes x 2 sets x 8 ranks would be 57 GB of host memory)."""
        rng = np.random.default_rng(1234 + rank)
        vols = []
        for _ in range(2):
            v = torch.empty((batch, 1) + img, dtype=torch.float32, device=dev)
            for i in range(batch):
                v[i, 0].copy_(torch.from_numpy(rng.random(img, dtype=np.float32)))
            vols.append(v)
        slices = [torch.from_numpy(rng.standard_normal((batch, 32, 768), dtype=np.float32)).to(dev) for _ in range(2)]
        return vols, slices

    # The default batch takes ~200 GB of HBM.  If it does not fit (a GPU that is not empty), every rank falls back TOGETHER to the next
    # smaller candidate -- the batch that ran is in config.batch_per_gpu; an explicit --batch is never changed.
    reserve_gb = float(os.environ.get("HS_BENCH_RESERVE_GB", "0"))       # (test hook: occupy part of the HBM to exercise the fallback)
    reserved = torch.empty(int(reserve_gb * 1e9), dtype=torch.uint8, device=dev) if reserve_gb > 0 else None
    candidates = [a.batch] + ([b for b in (31, 20, 8) if b < a.batch] if batch_defaulted else [])
    for cand in candidates:
        eng = vols = slices = None
        ok = 1
        try:
            eng, pe, needs_2d = build_engine(a.workload, geom, cand, dev, precision=a.precision)
            vols, slices = inputs(cand)
            # one untimed forward + backward WITHOUT collectives: the engines size some workspaces on first use, and a rank that runs
            # out of memory must find out before the vote below, not inside a step whose all-reduces the other ranks are waiting in
            if hasattr(eng, "probe_step"):
                eng.probe_step(*((vols[0], slices[0]) if needs_2d else (vols[0],)))
            torch.cuda.synchronize()
        except torch.cuda.OutOfMemoryError:
            ok = 0
        if world > 1:
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            a.batch = cand
            break
        if rank == 0:
            print(f"[bench] {cand} volumes per GPU do not fit in HBM here; trying a smaller batch", file=sys.stderr, flush=True)
        del eng, vols, slices
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
    else:
        sys.exit("bench.py: no candidate batch fits in HBM")
    logit_factor = set_attn_logit_std(encoders_of(eng), a.attn_logit_std) if a.attn_logit_std else None
    dt, step = timed_run(eng, needs_2d, vols, slices, a.steps, a.warmup, sync)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(eng.flat).all().item())
    replicas_equal = None
    if world > 1:        # data-parallel replicas must hold the SAME parameters after the timed steps: a checksum's max and min over ranks
        chk = eng.flat.double().sum().reshape(1)
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        replicas_equal = bool((lo_ == hi_).item())

    # ---- instrumented pass: per-kernel-class HIP-event timing on the launch stream, same steps --------------
    # EVERY rank runs these steps (they contain the gradient all-reduces); only rank 0 records and reads the timers.
    def instrumented(step_fn, n):
        """-> (classes, roofline) of n more steps with the library's per-kernel-class timers on (rank 0; others ([], None))."""
        # per-kernel durations are only meaningful for kernels that run ALONE: an engine that forks its two encoder chains onto
        # two streams (DualEngine below 16 volumes per GPU) runs this pass on one stream
        forked = getattr(eng, "two_streams", False) and getattr(eng, "side", None) is not None
        if forked:
            side, eng.side = eng.side, None
        classes_, roof_ = [], None
        if rank == 0:
            L.call("hs_prof_enable", 1)
        for i in range(n):
            step_fn(i)
        torch.cuda.synchronize()
        if rank == 0:
            classes_ = profile_classes()
            L.call("hs_prof_enable", 0)
            for c in classes_:
                c["avg_us"] = 1e3 * c["total_ms"] / c["launches"]
                c["tflops"] = c["flops"] / (c["total_ms"] * 1e-3) / 1e12 if c["flops"] else 0.0
                c["gbs"] = c["bytes"] / (c["total_ms"] * 1e-3) / 1e9
            mf = [c for c in classes_ if c["flops"]]
            dom = max(mf, key=lambda c: c["total_ms"])
            pmc = measured_pmc(dom["name"], a.batch, a.workload, L.build_id())
            roof_ = dict(kernel=dom["name"], bound="mfma", achieved=round(dom["tflops"], 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                         frac=round(dom["tflops"] / PEAK_BF16_TFLOPS, 4), traffic=pmc.get("hbm_bytes_per_launch"),
                         avg_launch_us=round(dom["avg_us"], 2), flops_per_launch=dom["flops"] / dom["launches"])
            if pmc.get("stale"):
                roof_["traffic_note"] = (f"{pmc['stale']} was collected with library build {pmc['build_id']}, this run loaded {L.build_id()}: "
                                         "its counters are not reported (re-run tools/r6_profiles.sh)")
            elif pmc:
                roof_["traffic_source"], roof_["traffic_build_id"] = pmc["source"], pmc["build_id"]
                if pmc.get("mfma_busy_frac") is not None:
                    roof_["mfma_busy_frac_pmc"] = round(pmc["mfma_busy_frac"], 4)
            if forked:
                roof_["note"] = "instrumented pass on one stream (the timed pass forks the two encoder chains onto two streams)"
        if forked:
            eng.side = side
        if world > 1:
            dist.barrier()
        return classes_, roof_

    n_inst = max(1, min(a.steps, 3))
    roof, classes = None, []
    if not a.no_roofline:
        classes, roof = instrumented(step, n_inst)

    # ---- the north-star's side figures: optimizer excluded, peaky attention ------------------------------------------------
    extras = {}
    if not a.no_extras and a.workload != "mrg_c5":
        k = max(2, min(a.steps, 5))
        dtn = timed_run(eng, needs_2d, vols, slices, k, 1, sync, optimizer=False)[0]       # every rank: the all-reduces are inside
        if world > 1:
            t = torch.tensor([dtn], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtn = float(t.item())
        extras["optimizer_excluded"] = {"value": round(world * a.batch * k / dtn, 3), "unit": "CT-volumes/sec", "ms_per_step": round(1e3 * dtn / k, 3),
                                        "steps": k, "what": "forward + backward + gradient all-reduce, no AdamW / clip pass"}
        if world > 1 and hasattr(eng, "buckets"):
            # what the gradient exchange costs THIS step: the same forward + backward with the collectives muted (no optimizer pass in
            # either, so the replicas stay identical), timed the same way -- the difference is the communication the backward did not hide
            bk = eng.buckets
            muted_engines = [eng] + ([eng.s1] if getattr(eng, "s1", None) is not None else [])
            for e_ in muted_engines:
                e_.buckets.muted = True
            try:
                dtm = timed_run(eng, needs_2d, vols, slices, k, 1, sync, optimizer=False)[0]
            finally:
                for e_ in muted_engines:
                    e_.buckets.muted = False
            t = torch.tensor([dtm], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = float(t.item())
            wire = 2 if bk.wire == "bf16" else 4
            extras["comm"] = {"comm_bytes_per_step": int(sum((b_ - a_) * wire for e_ in muted_engines for a_, b_ in e_.buckets.bounds)),
                              "buckets": sum(len(e_.buckets.bounds) for e_ in muted_engines), "wire": bk.wire, "algo": bk.algo,
                              "comm_ms_exposed": round(1e3 * (dtn - dtm) / k, 3), "ms_per_step_muted": round(1e3 * dtm / k, 3),
                              "what": "gradient bytes each rank puts on the wire per step (flat fp32 gradient in backward-completion-order "
                                      "buckets) and the part of their exchange the backward did not hide: (fwd + bwd + exchange) - (fwd + bwd "
                                      "with the collectives muted), max over ranks"}
        if world == 1 and a.attn_logit_std is None and encoders_of(eng) and not a.no_roofline:
            f = set_attn_logit_std(encoders_of(eng), 1.5)
            dtp, stepp = timed_run(eng, needs_2d, vols, slices, k, 1, sync)
            cls_p, roof_p = instrumented(stepp, n_inst)
            stepp = None                                  # (the closure holds the engine: its HBM must be free for the f16_train leg)
            extras["peaky_attention"] = {"attn_logit_std": 1.5, "q_weight_factor": round(f, 3), "value": round(a.batch * k / dtp, 3),
                                         "unit": "CT-volumes/sec", "ms_per_step": round(1e3 * dtp / k, 3), "steps": k,
                                         "attention": attention_aggregate(cls_p),
                                         "roofline": {kk: roof_p[kk] for kk in ("kernel", "achieved", "frac", "avg_launch_us")} if roof_p else None,
                                         "what": "the same step after scaling the q rows of every qkv weight so that the scaled logits have std "
                                                 "1.5 (random init: 0.31, near-uniform attention; trained attention is peaky and the attention "
                                                 "kernels draw more power on such data)"}

    Np, Pd = pe.n_patches, pe.patch_dim
    fb = flops_per_volume(a.workload, Np, Pd)
    vols_per_s = world * a.batch * a.steps / dt
    desc = {
        "dual256": "dual 3D-ViT + 2E3-ViT encoders (patch {p}, 12 layers, d=768, N={N} tokens each) + two Spatial Packers -> (B,256,3072), "
                   "loss mean-square, fwd+bwd+grad all-reduce+clipped AdamW",
        "native_dual": "dual 3D-ViT + 2E3-ViT encoders (patch {p}, 12 layers, d=768, N={N} tokens each) + two Spatial Packers -> (B,256,3072), "
                       "loss mean-square, fwd+bwd+grad all-reduce+clipped AdamW",
        "vit256": "3D-ViT patch {p} 12 layers d=768 (N={N} tokens), fwd+bwd+grad all-reduce+clipped AdamW",
        "clip_c4a": "stage-2 CLIP step (CLIP_stage2.py:119-151): frozen 3D-ViT forward + 2E3-ViT fwd+bwd (patch {p}, N={N}), cls -> "
                    "mm_vision_proj -> normalize, all-gather, InfoNCE vs frozen text embeds + relation MSE, grad all-reduce+clipped AdamW",
        "clip_c4b": "both-trainable CLIP step (SURVEY 8(d) C4-B): the stage-1 step (CLIP_stage1.py:111-139: 3D-ViT fwd+bwd, InfoNCE) and the stage-2 "
                    "step (CLIP_stage2.py:119-151: 2E3-ViT fwd+bwd, InfoNCE + relation MSE against the detached stage-1 logits) on one volume batch "
                    "(patch {p}, N={N}), one shared stage-1 forward and feature all-gather, two grad all-reduces + clipped AdamW states",
        "clip_s1": "stage-1 CLIP step (CLIP_stage1.py:111-139): 3D-ViT fwd+bwd (patch {p}, N={N}), cls -> mm_vision_proj -> normalize, all-gather, "
                   "InfoNCE vs frozen text embeds, grad all-reduce+clipped AdamW",
        "mrg_c5": "MRG finetune step: frozen dual encoders (patch {p}, N={N}) forward, two Spatial Packers fwd+bwd, 256 image tokens spliced "
                  "into 800 token embeddings of a random-init frozen Phi-4-mini-shaped HF decoder (PyTorch-ROCm, out of the HIP path's scope; its "
                  "FLOPs are NOT in model_tflops), LM loss, packer grad all-reduce+clipped AdamW",
    }[a.workload].format(p=patch, N=Np + 1)
    out = {
        "metric": METRIC, "value": round(vols_per_s, 3), "unit": "CT-volumes/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": f"BASELINE {geom['cfg']}: {img[0]}x{img[1]}x{img[2]} volume, {desc}, random-init weights",
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                   "grad_exchange": f"{os.environ.get('HS_GRAD_WIRE', 'fp32')}/{os.environ.get('HS_GRAD_ALGO', 'allreduce')}"},
        "model_tflops": round(vols_per_s * fb / 1e12, 2),
        "mfma_roofline_frac_end_to_end": round(vols_per_s * fb / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "outputs_finite": finite,
        **({"replicas_equal": replicas_equal} if replicas_equal is not None else {}),
        "peak_hbm_gb": round((torch.cuda.max_memory_allocated(dev) - (reserved.numel() if reserved is not None else 0)) / 1e9, 2),
    }
    if os.environ.get("HS_RECOMPUTE_MLP") and a.workload in ("dual256", "native_dual"):
        out["config"]["recompute"] = "mlp activations re-made in the backward (one more fc1 + GELU GEMM per layer)"
    if a.attn_logit_std:
        out["config"]["attn_logit_std"] = a.attn_logit_std
        out["config"]["q_weight_factor"] = round(logit_factor, 3) if logit_factor else None
    if roof:
        out["roofline"] = roof
        att = attention_aggregate(classes)
        if att:
            att["ms_per_step_share"] = round(att["total_ms"] / n_inst / (1e3 * dt / a.steps), 4)
            out["attention"] = att
            attf = attention_aggregate(classes, "attn_fwd")
            if attf:
                attf["ms_per_step_share"] = round(attf["total_ms"] / n_inst / (1e3 * dt / a.steps), 4)
                out["attention_fwd"] = attf
        out["kernels"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items() if k != "bytes" and k != "flops"}
                          for c in sorted(classes, key=lambda c: -c["total_ms"])]

    out.update(extras)

    # ---- PCIe-inclusive rate: the volumes start in (pinned) HOST memory and are uploaded for every step --------------------------------------
    if not a.no_extras and world == 1 and a.workload in ("dual256", "native_dual", "vit256"):
        try:
            extras_h2d = pcie_inclusive_rate(eng, needs_2d, vols, slices, dev, sync, max(3, min(a.steps, 5)))
            if extras_h2d:
                out["pcie_inclusive"] = extras_h2d
        except RuntimeError as e:                  # (no pinned host memory to spare: report nothing rather than fail the bench)
            print(f"[bench] pcie_inclusive leg skipped: {e}", file=sys.stderr, flush=True)

    # ---- the same step with fp16 operands (round 5): the mode whose outputs AND gradients sit inside ~1e-3 of the fp32 reference -----------
    if not a.no_extras and world == 1 and a.precision == "bf16" and a.workload in ("dual256", "native_dual", "vit256"):
        del eng, step
        torch.cuda.empty_cache()
        eng16, _, _ = build_engine(a.workload, geom, a.batch, dev, precision="f16")
        k16 = max(2, min(a.steps, 5))
        dt16 = timed_run(eng16, needs_2d, vols, slices, k16, 2, sync)[0]      # ([0]: the returned step closure would keep the engine alive)
        fin16 = bool(torch.isfinite(eng16.flat).all().item()) and bool(torch.isfinite(eng16.params.gsq).all().item())
        out["f16_train"] = {"value": round(a.batch * k16 / dt16, 3), "unit": "CT-volumes/sec", "ms_per_step": round(1e3 * dt16 / k16, 3), "steps": k16,
                            "vs_bf16": round((a.batch * k16 / dt16) / vols_per_s, 4), "loss_scale": eng16.loss_scale, "outputs_finite": fin16,
                            "steps_applied": eng16.step_no, "steps_skipped": eng16.skipped_steps,
                            "what": "the SAME step (same kernels, batch, optimizer) with IEEE fp16 in place of bf16 in every 16-bit tensor, "
                                    "forward and backward, power-of-two loss scale taken out inside AdamW (engine precision='f16'); "
                                    "steps_skipped = optimizer passes the device refused for a non-finite gradient norm (must be 0)"}
        del eng16
        eng = None
        torch.cuda.empty_cache()
        acc = train_accuracy_vs_f32(a.workload, geom, dev)
        if acc:
            out["f16_train"]["accuracy"] = acc
        torch.cuda.empty_cache()

    # ---- the same step at the per-GPU batches the reference itself trains with: 1 volume (configs[3]/[4]: batch 8 over 8 GPUs), 2 (MRG
    # finetune, script/train_vlm_mrg.sh:15) and 3 (CLIP pretrain, script/train_clip_stage1.sh:12) ----------
    engine_rates = {}
    if not a.no_batch1 and a.batch != 1 and world == 1:
        eng = step = None
        del vols, slices
        torch.cuda.empty_cache()
        what = {1: "1 volume per GPU = BASELINE configs[3] / [4] (batch 8 over 8 GPUs)",
                2: "2 volumes per GPU = the reference's MRG-finetuning per-device batch (script/train_vlm_mrg.sh:15)",
                3: "3 volumes per GPU = the reference's CLIP-pretraining per-device batch (script/train_clip_stage1.sh:12)"}
        for bsmall in (1, 2, 3):
            if bsmall >= a.batch:
                continue
            engs, _, _ = build_engine(a.workload, geom, bsmall, dev, precision=a.precision)
            vs_, ss_ = inputs(bsmall)
            ns = max(5, a.steps, 36 // bsmall)                    # ~0.5 s of timed work: a one-volume step is 15 ms
            dts = timed_run(engs, needs_2d, vs_, ss_, ns, max(3, a.warmup), sync)[0]
            out[f"batch{bsmall}"] = {"value": round(bsmall * ns / dts, 3), "unit": "CT-volumes/sec", "ms_per_step": round(1e3 * dts / ns, 3), "steps": ns,
                                     "batch_per_gpu": bsmall, "what": what[bsmall],
                                     "mfma_roofline_frac_end_to_end": round(bsmall * ns / dts * fb / 1e12 / PEAK_BF16_TFLOPS, 4)}
            engine_rates[(bsmall, a.precision)] = bsmall * ns / dts
            del engs, vs_, ss_
            torch.cuda.empty_cache()
    # ---- the step a DROP-IN user runs: the reference's own call pattern (nn.Module.forward -> autograd -> torch.optim.AdamW + clip) on the
    # hsenet_amd modules, beside the engine's rate on the same box / batch / inputs (bench_module_api.py) ----------
    if not a.no_extras and world == 1 and a.workload == "dual256" and not a.no_module_api:
        eng = vols = slices = step = None
        torch.cuda.empty_cache()
        from bench_module_api import module_api_train
        mat = module_api_train(dev, batches=(1, 3, 13), precisions=("bf16", "f16"), steps=6, warmup=3, geom=geom, engine_rates=engine_rates)
        out["module_api_train"] = mat
        # the batch sweep as one object (volumes/s of the ENGINE step at each per-GPU batch; 13 from the leg above, the headline batch last)
        sweep = {str(b): out[f"batch{b}"]["value"] for b in (1, 2, 3) if f"batch{b}" in out}
        if a.precision in mat.get("b13", {}):
            sweep["13"] = mat["b13"][a.precision]["engine"]
        sweep[str(a.batch)] = round(vols_per_s, 3)
        out["batch_sweep"] = {"unit": "CT-volumes/sec", "volumes_per_gpu": sweep,
                              "what": "engine step (fwd + bwd + clipped AdamW) at each per-GPU batch; 13 volumes = 51 GB of HBM, the headline's 53 = 197 GB"}
    if rank == 0 and world == 1 and not a.no_extras:
        eng = vols = slices = step = None
        torch.cuda.empty_cache()
        f32 = f32_mode_rate(a.workload, geom, dev)
        if f32:
            out["f32_mode"] = f32
        torch.cuda.empty_cache()
        ev = eval_forward_rates(a.workload, geom, dev)
        if ev:
            out["eval_forward"] = ev
        torch.cuda.empty_cache()
        el = eval_latency(a.workload, geom, dev)
        if el:
            out["eval_latency_ms"] = el
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload != "mrg_c5":      # (the LLM-dominated step has no CPU leg)
        out["cpu_baseline"] = cpu_baseline(a.workload, geom)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
