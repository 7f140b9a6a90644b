#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE modules (build container only).

TEST INFRASTRUCTURE.  Reads /root/reference at run time (never copies from it):
  * multimodal_projector/spatial_pooling_projector.py is imported as-is;
  * multimodal_encoder/vit.py is imported with ``sys.modules`` stubs for ``monai`` (its two block classes
    are bound to the oracle's restatement — MONAI 1.3.0 is absent, see ref_cpu.py header) and ``open_clip``.
For each case the oracle restatement and the reference get the same seeded weights/inputs
(``ref_cpu.seeded_state_dict`` / ``synthetic_*``), are asserted equal, and the REFERENCE's outputs
(or a row subset + per-token norms for the big geometries) are written as fixtures.  Weights and
inputs are NOT stored: tests regenerate them from the seeds recorded in each file.

Usage:  python oracle/make_goldens.py [--skip-big] [--only grads]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R  # noqa: E402

REF = "/root/reference/Preprint/LaMed/src/model"
OUT = os.path.join(ROOT, "tests", "golden")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for n in ("monai", "monai.networks", "monai.networks.blocks", "monai.networks.blocks.patchembedding",
              "monai.networks.blocks.transformerblock", "open_clip"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["monai.networks.blocks.patchembedding"].PatchEmbeddingBlock = R.PatchEmbeddingBlock
    sys.modules["monai.networks.blocks.transformerblock"].TransformerBlock = R.TransformerBlock
    vit = _load(os.path.join(REF, "multimodal_encoder", "vit.py"), "_ref_vit")
    spp = _load(os.path.join(REF, "multimodal_projector", "spatial_pooling_projector.py"), "_ref_spp")
    return vit, spp


def _eq(a, b, what, tol=1e-6):
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert err <= tol, f"restatement != reference for {what}: {err}"
    return err


def _sel(t, rows):
    return t[:, rows].contiguous().numpy()


class Cfg:
    vision_select_layer = -1
    vision_select_feature = "patch"
    image_channel = 1
    remain_2d3d_ViT_type = "dual_vits"

    def __init__(self, image_size, patch_size):
        self.image_size, self.patch_size = image_size, patch_size


@torch.no_grad()
def tiny_encoders(vit):
    """(i) tiny geometry, FULL tensors incl. every hidden state: img (8,32,32), patch (4,16,16) -> 8 tokens."""
    img, patch, L, B = (8, 32, 32), (4, 16, 16), 2, 2
    kw = dict(in_channels=1, img_size=img, patch_size=patch, num_layers=L, pos_embed="perceptron",
              classification=True)
    out = dict(img=np.array(img), patch=np.array(patch), layers=L, batch=B, seed_w1=11, seed_w2=12, seed_x=13,
               seed_s=14)
    x = R.synthetic_volume((B, 1) + img, 13)
    s = R.synthetic_slices(B, 14)
    for tag, cls_ref, cls_ora, seed in (("s1", vit.ViT_stage1, R.ViT_stage1, 11), ("s2", vit.ViT_stage2, R.ViT_stage2, 12)):
        ref, ora = cls_ref(**kw).eval(), cls_ora(**kw).eval()
        assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
        sd = R.seeded_state_dict(ora, seed)
        ref.load_state_dict(sd), ora.load_state_dict(sd)
        args = (x,) if tag == "s1" else (x, s)
        yr, hr = ref(*args)
        yo, ho = ora(*args)
        _eq(yo, yr, f"tiny {tag} out")
        for a, b in zip(ho, hr):
            _eq(a, b, f"tiny {tag} hidden")
        out[f"{tag}_out"] = yr.numpy()
        out[f"{tag}_hidden"] = torch.stack(hr).numpy()
        out[f"{tag}_keys"] = np.array(list(ref.state_dict().keys()))
    # regular_attention alone (P5) with its attention map
    ra_ref, ra_ora = vit.regular_attention().eval(), R.regular_attention().eval()
    sd = R.seeded_state_dict(ra_ora, 15)
    ra_ref.load_state_dict(sd), ra_ora.load_state_dict(sd)
    q = torch.from_numpy(np.random.default_rng(16).standard_normal((B, 8, 768), dtype=np.float32))
    (yr, ar), (yo, ao) = ra_ref(q, s, s), ra_ora(q, s, s)
    _eq(yo, yr, "regular_attention"), _eq(ao, ar, "regular_attention map")
    out.update(ra_out=yr.numpy(), ra_map=ar.numpy(), seed_ra=15, seed_raq=16)
    np.savez_compressed(os.path.join(OUT, "tiny_encoders.npz"), **out)
    print("tiny_encoders ok")


@torch.no_grad()
def packer(spp):
    """(ii) Spatial Packer: resolution_attention_v3 full tensors on small grids (reference forward is
    grid-agnostic) + the wrapper at its hard-coded native grid (row subset + norms)."""
    out = {}
    ra_ref, ra_ora = spp.resolution_attention_v3().eval(), R.resolution_attention_v3().eval()
    sd = R.seeded_state_dict(ra_ora, 21)
    ra_ref.load_state_dict(sd), ra_ora.load_state_dict(sd)
    for tag, grid, ks in (("g288_k144", (2, 8, 8), (1, 4, 4)), ("g444_k222", (4, 4, 4), (2, 2, 2)),
                          ("g488_k244", (4, 8, 8), (2, 4, 4))):
        hr = torch.from_numpy(np.random.default_rng(22).standard_normal((2,) + grid + (768,), dtype=np.float32))
        lr = torch.nn.functional.avg_pool3d(hr.permute(0, 4, 1, 2, 3), kernel_size=ks).permute(0, 2, 3, 4, 1)
        yr, yo = ra_ref(lr, hr, ks), ra_ora(lr, hr, ks)
        _eq(yo, yr, f"resolution_attention_v3 {tag}")
        out[f"ra_{tag}"] = yr.numpy()
    out.update(seed_ra=21, seed_hr=22)
    # wrapper, native geometry; the survey's known answer is re-checked here
    args = ((32, 256, 256), (4, 16, 16), 768, 3072, "mlp", 2)
    torch.manual_seed(0)
    ref = spp.VisualPacker_3d_phi_v3(*args).eval()
    xs = torch.randn(2, 2048, 768)
    ka = float(ref(xs).abs().mean())
    assert abs(ka - 0.15821750462055206) < 1e-6, ka
    ora = R.VisualPacker_3d_phi_v3(*args).eval()
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
    sd = R.seeded_state_dict(ora, 23)
    ref.load_state_dict(sd), ora.load_state_dict(sd)
    x = torch.from_numpy(np.random.default_rng(24).standard_normal((2, 2048, 768), dtype=np.float32))
    yr, yo = ref(x), ora(x)
    _eq(yo, yr, "VisualPacker_3d_phi_v3 native")
    assert ref.proj_out_num == ora.proj_out_num == 128
    rows = [0, 1, 63, 127]
    out.update(packer_rows=np.array(rows), packer_out_rows=_sel(yr, rows), packer_row_norms=yr.norm(dim=-1).numpy(),
               packer_keys=np.array(list(ref.state_dict().keys())), seed_pw=23, seed_px=24, packer_mean_abs=ka)
    np.savez_compressed(os.path.join(OUT, "packer.npz"), **out)
    print("packer ok")


@torch.no_grad()
def big(vit, spp):
    """(iii) reference-native 32x256x256 / (4,16,16), L=12 dual tower + packers; (iv) 256^3 / 16^3 stage-1."""
    out = {}
    img, patch = (32, 256, 256), (4, 16, 16)
    ref, ora = vit.ViT3DTower_dual_encoders(Cfg(img, patch)).eval(), R.ViT3DTower_dual_encoders(Cfg(img, patch)).eval()
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
    sd = R.seeded_state_dict(ora, 31)
    ref.load_state_dict(sd), ora.load_state_dict(sd)
    x, s = R.synthetic_volume((1, 1) + img, 32), R.synthetic_slices(1, 33)
    (r1, r2), (o1, o2) = ref(x, s), ora(x, s)
    _eq(o1, r1, "native stage1"), _eq(o2, r2, "native stage2")
    rows = [0, 1, 2, 1024, 2046, 2047]          # patch-token rows (cls already dropped by the tower)
    out.update(native_rows=np.array(rows), native_s1_rows=_sel(r1, rows), native_s2_rows=_sel(r2, rows),
               native_s1_norms=r1.norm(dim=-1).numpy(), native_s2_norms=r2.norm(dim=-1).numpy(),
               native_s1_stats=np.array([float(r1.mean()), float(r1.std())]),
               native_s2_stats=np.array([float(r2.mean()), float(r2.std())]),
               seed_tower=31, seed_x=32, seed_s=33, n_tower_keys=len(sd))
    # cls rows come from the stage modules directly (the tower drops them)
    c1, _ = ref.vision_tower_stage1(x)
    c2, _ = ref.vision_tower_stage2(x, s)
    out.update(native_s1_cls=c1[:, 0].numpy(), native_s2_cls=c2[:, 0].numpy())
    # P8: two packers on the tower output -> (1,256,3072)
    pk_args = (img, patch, 768, 3072, "mlp", 2)
    p1r, p2r = spp.VisualPacker_3d_phi_v3(*pk_args).eval(), spp.VisualPacker_3d_phi_v3(*pk_args).eval()
    p1o, p2o = R.VisualPacker_3d_phi_v3(*pk_args).eval(), R.VisualPacker_3d_phi_v3(*pk_args).eval()
    for (pr, po, seed) in ((p1r, p1o, 34), (p2r, p2o, 35)):
        sdp = R.seeded_state_dict(po, seed)
        pr.load_state_dict(sdp), po.load_state_dict(sdp)
    yr = torch.cat([p1r(r1), p2r(r2)], dim=1)       # lamed_arch.py:127-132
    yo = R.encode_images(ora, p1o, p2o, x, s)
    _eq(yo, yr, "encode_images native")
    prow = [0, 1, 127, 128, 255]
    out.update(enc_rows=np.array(prow), enc_out_rows=_sel(yr, prow), enc_row_norms=yr.norm(dim=-1).numpy(),
               seed_p1=34, seed_p2=35)
    np.savez_compressed(os.path.join(OUT, "native_dual.npz"), **out)
    print("native_dual ok")

    out = {}
    img, patch = (256, 256, 256), (16, 16, 16)
    kw = dict(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True)
    ref, ora = vit.ViT_stage1(**kw).eval(), R.ViT_stage1(**kw).eval()
    sd = R.seeded_state_dict(ora, 41)
    ref.load_state_dict(sd), ora.load_state_dict(sd)
    x = R.synthetic_volume((1, 1) + img, 42)
    (yr, _), (yo, _) = ref(x), ora(x)
    _eq(yo, yr, "256^3 stage1")
    rows = [0, 1, 2, 513, 1024, 2049, 3072, 4095, 4096]   # token rows incl. cls (row 0)
    out.update(rows=np.array(rows), out_rows=_sel(yr, rows), norms=yr.norm(dim=-1).numpy(),
               stats=np.array([float(yr.mean()), float(yr.std())]), seed_w=41, seed_x=42)
    np.savez_compressed(os.path.join(OUT, "vit256_stage1.npz"), **out)
    print("vit256_stage1 ok")


# --------------------------------------------------------------------------------------
# gradient fixtures (round 2): autograd of sum(out * w) through the REFERENCE modules
# --------------------------------------------------------------------------------------
FULL_BLOCKS = (0, 5, 11)      # blocks whose gradients are stored as row subsets; every tensor gets its norm


def _grad_fixture(named, tag):
    """{tag}_names, {tag}_norms for every tensor; for small tensors and the tensors of blocks 0/5/11, the embeddings, the scorer
    and the packers also values: full when numel <= 4096, else rows [0, mid, last] of the 2-D view (last dim kept)."""
    names, norms, vals = [], [], {}
    for n, p in named:
        g = p.grad
        assert g is not None, n
        names.append(n)
        norms.append(float(g.double().norm()))
        blk = None
        parts = n.split(".")
        if "blocks" in parts:
            blk = int(parts[parts.index("blocks") + 1])
        if blk is not None and blk not in FULL_BLOCKS:
            continue
        g2 = g.reshape(-1, g.shape[-1]) if g.dim() >= 2 else g.reshape(1, -1)
        if g.numel() <= 4096:
            vals[f"{tag}_g/{n}"] = g.numpy().copy()
        else:
            rows = sorted({0, g2.shape[0] // 2, g2.shape[0] - 1})
            vals[f"{tag}_g/{n}"] = g2[rows].numpy().copy()
            vals[f"{tag}_r/{n}"] = np.array(rows)
    vals[f"{tag}_names"] = np.array(names)
    vals[f"{tag}_norms"] = np.array(norms)
    return vals


def _grads_equal(named_a, named_b, what, tol=2e-5):
    worst = 0.0
    for (na, pa), (nb, pb) in zip(named_a, named_b):
        assert na == nb
        den = float(pb.grad.double().norm())
        err = float((pa.grad.double() - pb.grad.double()).norm()) / max(den, 1e-30)
        if den > 1e-9:
            worst = max(worst, err)
            assert err <= tol, f"restatement gradient != reference gradient for {what}: {na} {err}"
    return worst


def _packer_apply(pk, feats, grid, kernel):
    """VisualPacker_3d_phi_v3.forward (spatial_pooling_projector.py:138-146) with the token grid / pooling kernel as arguments
    (the reference hard-codes (8,16,16) / (1,4,4)); the arithmetic is the reference module's own resolution_attention + proj_mpls."""
    b = feats.size(0)
    hr = feats.view(b, *grid, 768)
    lr = torch.nn.functional.avg_pool3d(hr.clone().permute(0, 4, 1, 2, 3), kernel_size=kernel).permute(0, 2, 3, 4, 1)
    return pk.proj_mpls(pk.resolution_attention(lr, hr, kernel)).view(b, -1, pk.out_dim)


def _dual_case(vit, spp, img, patch, grid, kernel, seeds, use_wrapper):
    """fwd + bwd of encode_images through the reference tower and two reference packers; returns (fixture dict, out)."""
    s_t, s_p1, s_p2, s_x, s_s, s_w = seeds
    pk_args = (img, patch, 768, 3072, "mlp", 2)
    x, s = R.synthetic_volume((1, 1) + img, s_x), R.synthetic_slices(1, s_s)
    w = torch.from_numpy(np.random.default_rng(s_w).standard_normal((1, 256, 3072), dtype=np.float32))

    def run(tower_cls, packer_cls, reference):
        tower = tower_cls(Cfg(img, patch)).eval()
        p1, p2 = packer_cls(*pk_args).eval(), packer_cls(*pk_args).eval()
        for m, sd_seed in ((tower, s_t), (p1, s_p1), (p2, s_p2)):
            m.load_state_dict(R.seeded_state_dict(m, sd_seed))
        f1, f2 = tower(x, s)                                                           # vit.py:926-948
        if reference and use_wrapper:
            out = torch.cat([p1(f1), p2(f2)], dim=1)                                    # lamed_arch.py:127-132
        else:
            out = torch.cat([_packer_apply(p1, f1, grid, kernel), _packer_apply(p2, f2, grid, kernel)], dim=1)
        (out * w).sum().backward()
        named = ([("vision_tower." + n, p) for n, p in tower.named_parameters()] + [("mm_projector." + n, p) for n, p in p1.named_parameters()]
                 + [("mm_projector2." + n, p) for n, p in p2.named_parameters()])
        return out.detach(), named

    out_r, named_r = run(vit.ViT3DTower_dual_encoders, spp.VisualPacker_3d_phi_v3, True)
    fx = _grad_fixture(named_r, "grad")
    out_o, named_o = run(R.ViT3DTower_dual_encoders, R.VisualPacker_3d_phi_v3, False)
    _eq(out_o, out_r, f"encode_images {img}")
    worst = _grads_equal(named_o, named_r, f"dual {img}")
    prow = [0, 1, 127, 128, 255]
    fx.update(img=np.array(img), patch=np.array(patch), grid=np.array(grid), kernel=np.array(kernel), seed_tower=s_t, seed_p1=s_p1, seed_p2=s_p2,
              seed_x=s_x, seed_s=s_s, seed_w=s_w, out_rows_idx=np.array(prow), out_rows=_sel(out_r, prow),
              out_row_norms=out_r.norm(dim=-1).numpy())
    return fx, worst


def grads(vit, spp):
    """Gradient fixtures from the imported reference: loss = sum(out * w), w ~ N(0,1) seeded.
      grad_native_dual.npz : reference-native 32x256x256 dual tower + the reference packer WRAPPER -> (1,256,3072)   (configs[2] at
                             the geometry the reference hard-codes)
      grad_dual256.npz     : BASELINE configs[2], 256^3 / 16^3: reference tower + the reference packers' own resolution_attention /
                             proj_mpls on the (16,16,16) grid with kernel (2,4,4) (the wrapper's .view is the only line bypassed)
      grad_vit256_stage1.npz: BASELINE configs[1], 256^3 ViT_stage1 alone, loss on the (1,4097,768) output
    Each holds per-tensor gradient norms for EVERY parameter tensor and values (full or rows [0, mid, last]) for the embeddings,
    scorer, final norm, packers and blocks 0 / 5 / 11.  The restatement's gradients are asserted equal to the reference's."""
    fx, worst = _dual_case(vit, spp, (32, 256, 256), (4, 16, 16), (8, 16, 16), (1, 4, 4), (61, 62, 63, 64, 65, 66), True)
    np.savez_compressed(os.path.join(OUT, "grad_native_dual.npz"), **fx)
    print(f"grad_native_dual ok (restatement vs reference gradients: worst {worst:.1e})")
    fx, worst = _dual_case(vit, spp, (256, 256, 256), (16, 16, 16), (16, 16, 16), (2, 4, 4), (71, 72, 73, 74, 75, 76), False)
    np.savez_compressed(os.path.join(OUT, "grad_dual256.npz"), **fx)
    print(f"grad_dual256 ok (restatement vs reference gradients: worst {worst:.1e})")

    img, patch = (256, 256, 256), (16, 16, 16)
    kw = dict(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True)
    x = R.synthetic_volume((1, 1) + img, 82)
    w = torch.from_numpy(np.random.default_rng(83).standard_normal((1, 4097, 768), dtype=np.float32))
    res = []
    for cls in (vit.ViT_stage1, R.ViT_stage1):
        m = cls(**kw).eval()
        m.load_state_dict(R.seeded_state_dict(m, 81))
        y, _ = m(x)
        (y * w).sum().backward()
        res.append((y.detach(), list(m.named_parameters())))
    _eq(res[1][0], res[0][0], "256^3 stage1 (grad case)")
    worst = _grads_equal(res[1][1], res[0][1], "256^3 stage1")
    fx = _grad_fixture(res[0][1], "grad")
    fx.update(seed_w=81, seed_x=82, seed_dy=83)
    np.savez_compressed(os.path.join(OUT, "grad_vit256_stage1.npz"), **fx)
    print(f"grad_vit256_stage1 ok (restatement vs reference gradients: worst {worst:.1e})")


def clip_c4a():
    """BASELINE configs[3] at full geometry (256^3 / 16^3, L = 12), 2 volumes, global step 1000: loss pieces, logits and the gradient
    fixture of every trainable tensor of the stage-2 CLIP step, from the ORACLE's restatement (oracle/clip_cpu.py — CLIP_stage2.py
    itself cannot be imported, see that file's header: this fixture pins the HIP step to the oracle at full size, it does not pin
    the oracle's ~40 lines of loss glue to the reference; the encoders inside are the classes the fixtures above validate)."""
    from oracle import clip_cpu as RC
    img, patch, B, step = (256, 256, 256), (16, 16, 16), 2, 1000
    m = RC.M3DCLIPVision_stage2(img, patch).eval()
    sd = R.seeded_state_dict(m, 91)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)))
    sd["stage1_pretrained_CLIP.logit_scale"] = torch.tensor(float(np.log(1 / 0.07)))
    m.load_state_dict(sd)
    x, s = R.synthetic_volume((B, 1) + img, 92), R.synthetic_slices(B, 93)
    t1, t2 = RC.synthetic_text_features(B, 94), RC.synthetic_text_features(B, 95)
    out = m(x, t1, t2, s, global_step=step)
    out["loss"].backward()
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    fx = _grad_fixture(named, "grad")
    fx.update(img=np.array(img), patch=np.array(patch), batch=B, global_step=step, seed_w=91, seed_x=92, seed_s=93, seed_t1=94, seed_t2=95,
              loss=float(out["loss"]), loss_cl=float(out["loss_CL_stage2"]), loss_relation=float(out["loss_relation"]),
              logits=out["logits"].detach().numpy())
    np.savez_compressed(os.path.join(OUT, "clip_c4a_256.npz"), **fx)
    print(f"clip_c4a_256 ok: loss {float(out['loss']):.6f} (CL {float(out['loss_CL_stage2']):.6f}, relation {float(out['loss_relation']):.6f}), "
          f"{len(named)} trainable tensors")


@torch.no_grad()
def plumbing(vit):
    """BASELINE configs[0]: one 128^3 volume, ViT-small (6 layers, d=384, 6 heads, mlp 1536), fwd only."""
    img, patch = (128, 128, 128), (16, 16, 16)
    kw = dict(in_channels=1, img_size=img, patch_size=patch, hidden_size=384, mlp_dim=1536, num_layers=6,
              num_heads=6, pos_embed="perceptron", classification=True)
    ref, ora = vit.ViT_stage1(**kw).eval(), R.ViT_stage1(**kw).eval()
    sd = R.seeded_state_dict(ora, 51)
    ref.load_state_dict(sd), ora.load_state_dict(sd)
    x = R.synthetic_volume((1, 1) + img, 52)
    (yr, _), (yo, _) = ref(x), ora(x)
    _eq(yo, yr, "plumbing vit-small")
    np.savez_compressed(os.path.join(OUT, "plumbing_vit_small.npz"), out=yr.numpy(), seed_w=51, seed_x=52)
    print("plumbing ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-big", action="store_true")
    ap.add_argument("--only", default=None, choices=["grads", "clip"], help="regenerate only this group of fixtures")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present (this script only runs in the build container)")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    vit, spp = load_reference()
    if a.only == "grads":
        grads(vit, spp)
        return
    if a.only == "clip":
        clip_c4a()
        return
    tiny_encoders(vit)
    packer(spp)
    plumbing(vit)
    if not a.skip_big:
        big(vit, spp)
        grads(vit, spp)
        clip_c4a()


if __name__ == "__main__":
    main()
