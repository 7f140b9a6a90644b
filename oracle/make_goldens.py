#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE modules (build container only).

TEST INFRASTRUCTURE.  Reads /root/reference at run time (never copies from it):
  * multimodal_projector/spatial_pooling_projector.py is imported as-is;
  * multimodal_encoder/vit.py is imported with ``sys.modules`` stubs for ``monai`` (its two block classes
    are bound to the oracle's restatement — MONAI 1.3.0 is absent, see ref_cpu.py header) and ``open_clip``;
  * (round 5) the CLIP wrappers LaMed.src.model.CLIP_stage1 / CLIP_stage2 are imported as the reference's own package and RUN (random-init
    one-layer BERT in place of the absent pretrained weights; world size 1 and, in two gloo processes, world size 2); lamed_arch.py's
    LamedMetaForCausalLM mixin over the reference's builders (one stand-in module for the out-of-scope segmentation builder); the CT
    script's resize_array / nii_img_to_tensor on a synthetic scan.
For each case the oracle restatement and the reference get the same seeded weights/inputs
(``ref_cpu.seeded_state_dict`` / ``synthetic_*``), are asserted equal, and the REFERENCE's outputs
(or a row subset + per-token norms for the big geometries) are written as fixtures.  Weights and
inputs are NOT stored: tests regenerate them from the seeds recorded in each file.

Usage:  python oracle/make_goldens.py [--skip-big] [--only grads | clip | mixin | prep]
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


def check_rearrange_witness():
    """the reference records MONAI's patch rearrangement as a pasted module repr behind every PatchEmbeddingBlock call (vit.py:144, 299,
    437, 508, 788: `# Rearrange('b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)', p1=4, p2=16, p3=16) + Linear(in_features=1024, ...)`).
    Every one of them must be the string the oracle rearranges with, and the Linear's in_features must be p1*p2*p3*c: the layout of the
    hot path's first operand is then machine-checked against the only witness of MONAI's behaviour that exists in the reference tree."""
    import re
    text = open(os.path.join(REF, "multimodal_encoder", "vit.py")).read()
    hits = re.findall(r"#\s*Rearrange\('([^']+)',\s*p1=(\d+),\s*p2=(\d+),\s*p3=(\d+)\)\s*\+\s*Linear\(in_features=(\d+),\s*out_features=(\d+)", text)
    assert len(hits) >= 5, f"expected the pasted Rearrange repr behind every PatchEmbeddingBlock call, found {len(hits)}"
    for pat, p1, p2, p3, fin, fout in hits:
        assert pat == R.REARRANGE_PATTERN, f"reference comment records {pat!r}, the oracle rearranges with {R.REARRANGE_PATTERN!r}"
        assert int(p1) * int(p2) * int(p3) == int(fin) and int(fout) == 768, (p1, p2, p3, fin, fout)
    return len(hits)


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


# --------------------------------------------------------------------------------------
# the CLIP wrappers (round 5): M3DCLIP_stage1 / M3DCLIP_stage2 themselves, run here
# --------------------------------------------------------------------------------------
PREPRINT = "/root/reference/Preprint"
BERT_KW = dict(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=1536, vocab_size=30522)


def load_reference_clip():
    """Import LaMed.src.model.CLIP_stage1 / CLIP_stage2 as the reference's own package (its absolute imports need Preprint/ on the
    path).  Absent third-party pieces get stand-ins, as for vit.py: monai's two blocks = the oracle's restatement, open_clip = an
    empty module (only touched when config.use_mask and config.use_2D_Encoder), BertModel.from_pretrained = a random-init ONE-layer
    BertModel of the same width (no BERT weights exist offline; the text towers are out of scope and only their OUTPUTS are used:
    the reference's own encode_text results are what the oracle is fed)."""
    load_reference()
    if PREPRINT not in sys.path:
        sys.path.insert(0, PREPRINT)
    import transformers
    from transformers import BertConfig, BertModel

    def _bert(path, *a, **k):
        return BertModel(BertConfig(**BERT_KW))
    BertModel.from_pretrained = staticmethod(_bert)
    import LaMed.src.model.CLIP_stage1 as c1
    import LaMed.src.model.CLIP_stage2 as c2
    assert c1.BertModel is transformers.BertModel
    return c1, c2


def _ref_clip_models(c1, c2, img, patch, L, gather_loss, local_loss, seed_w, torch_seed=5):
    """(reference M3DCLIP_stage2 holding a reference M3DCLIP_stage1, oracle M3DCLIPVision_stage2) with identical seeded vision
    weights; BERT / mm_language_proj keep their torch.manual_seed init (they only produce the text features)."""
    from oracle import clip_cpu as RC
    torch.manual_seed(torch_seed)
    kw = dict(language_model_name_or_path="none", img_size=img, patch_size=patch, num_layers=L, gather_loss=gather_loss, local_loss=local_loss)
    s1 = c1.M3DCLIP_stage1(c1.M3DCLIPConfig_stage1(**kw)).eval()
    c2.AutoModel.from_pretrained = staticmethod(lambda path, **k: s1)                         # CLIP_stage2.py:87
    s2 = c2.M3DCLIP_stage2(c2.M3DCLIPConfig_stage2(pretrained_model="none", use_mask=False, use_2D_Encoder=False, **kw)).eval()
    ora = RC.M3DCLIPVision_stage2(img, patch, L, gather_loss=gather_loss, local_loss=local_loss).eval()
    sd = R.seeded_state_dict(ora, seed_w)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)))
    sd["stage1_pretrained_CLIP.logit_scale"] = torch.tensor(float(np.log(1 / 0.05)))          # differs: :128 must use the OUTER scale
    ora.load_state_dict(sd)
    missing, unexpected = s2.load_state_dict(sd, strict=False)
    assert not unexpected and all("language" in k for k in missing), (missing[:4], unexpected[:4])
    assert all(not p.requires_grad for p in s2.stage1_pretrained_CLIP.parameters())
    return s2, ora


def _tokens(B, seed):
    g = np.random.default_rng(seed)
    ids = torch.from_numpy(g.integers(1000, 30000, size=(B, 12)))
    mask = torch.ones(B, 12, dtype=torch.long)
    mask[:, 9:] = 0
    return ids, mask


def _clip_world_worker(rank, world, port, cases, q):
    """one rank of the reference at world size `world` (gloo): per case, this rank's loss / logits and its local vision gradients."""
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    c1, c2 = load_reference_clip()
    out = {}
    for tag, (stage, img, patch, L, B, step, gather_loss, local_loss, seeds) in cases.items():
        s2, _ = _ref_clip_models(c1, c2, img, patch, L, gather_loss, local_loss, seeds[0])
        x, s = R.synthetic_volume((B * world, 1) + img, seeds[1]), R.synthetic_slices(B * world, seeds[2])
        ids, mask = _tokens(B * world, seeds[3])
        sl = slice(rank * B, (rank + 1) * B)
        if not gather_loss:
            labels = torch.arange(B)
        elif local_loss:
            labels = rank * B + torch.arange(B)
        else:
            labels = torch.arange(B * world)
        if stage == 1:
            s2 = s2.stage1_pretrained_CLIP
            s2.requires_grad_(True)
            r = s2(x[sl], ids[sl], mask[sl], labels, global_step=step)
        else:
            r = s2(x[sl], ids[sl], mask[sl], labels, s[sl], global_step=step)
        r["loss"].backward()
        grads = {n: p.grad.numpy().copy() for n, p in s2.named_parameters() if p.requires_grad and p.grad is not None and "language" not in n}
        out[tag] = dict(loss=float(r["loss"].detach()), logits=r["logits"].detach().numpy().copy(), grads=grads)     # numpy: pickled by value
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def clip_ref():
    """tests/golden/clip_ref.npz: the reference's M3DCLIP_stage1.forward (CLIP_stage1.py:111-155) and M3DCLIP_stage2.forward
    (CLIP_stage2.py:119-190) at a small geometry, every branch; oracle/clip_cpu.py is asserted equal to each before it is written."""
    import torch.distributed as dist
    import torch.multiprocessing as mp
    from oracle import clip_cpu as RC
    c1, c2 = load_reference_clip()
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)   # gather_features needs a group (train_CLIP_stage1.py:31-38)
    img, patch, L, B = (8, 64, 64), (4, 16, 16), 2, 4
    seeds = dict(seed_w=101, seed_x=102, seed_s=103, seed_tok=104)
    x, s = R.synthetic_volume((B, 1) + img, 102), R.synthetic_slices(B, 103)
    ids, mask = _tokens(B, 104)
    labels = torch.arange(B)
    fx = dict(img=np.array(img), patch=np.array(patch), layers=L, batch=B, **seeds, scale_outer=float(np.log(1 / 0.07)),
              scale_inner=float(np.log(1 / 0.05)))
    worst = 0.0
    for gather_loss, local_loss, ftag in ((True, False, "g"), (True, True, "l"), (False, False, "n")):
        s2, ora = _ref_clip_models(c1, c2, img, patch, L, gather_loss, local_loss, 101)
        with torch.no_grad():
            t1 = s2.stage1_pretrained_CLIP.encode_text(ids, mask)[0][:, 0].clone()
            t2 = s2.encode_text(ids, mask)[0][:, 0].clone()
        if ftag == "g":
            fx.update(text1=t1.numpy(), text2=t2.numpy(), stage2_keys=np.array(list(s2.state_dict().keys())),
                      stage1_keys=np.array(list(s2.stage1_pretrained_CLIP.state_dict().keys())),
                      stage2_shapes=np.array([str(tuple(v.shape)) for v in s2.state_dict().values()]))
        # ---- stage 1, all three branches (its own logit_scale and flags; CLIP_stage1.py:111-155)
        s1, o1 = s2.stage1_pretrained_CLIP, ora.stage1_pretrained_CLIP
        s1.local_loss, s1.gather_loss, o1.local_loss, o1.gather_loss = local_loss, gather_loss, local_loss, gather_loss
        s1.requires_grad_(True), o1.requires_grad_(True)
        rr = s1(x, ids, mask, labels, global_step=500)
        ro = o1(x, t1, global_step=500)
        worst = max(worst, _eq(ro["loss"].detach(), rr["loss"].detach(), f"stage-1 loss {ftag}"),
                    _eq(ro["logits"].detach(), rr["logits"].detach(), f"stage-1 logits {ftag}"))
        rr["loss"].backward(), ro["loss"].backward()
        named_r = [(n, p) for n, p in s1.named_parameters() if "language" not in n]
        worst = max(worst, _grads_equal(list(o1.named_parameters()), named_r, f"stage-1 CLIP step {ftag}"))
        fx.update(_grad_fixture(named_r, f"s1{ftag}grad"))
        fx[f"s1_{ftag}_loss"], fx[f"s1_{ftag}_logits"] = float(rr["loss"].detach()), rr["logits"].detach().numpy()
        s1.requires_grad_(False), o1.requires_grad_(False)
        s1.zero_grad(), o1.zero_grad()
        if ftag != "g":
            # CLIP_stage2.py:124-128: the stage-1 features are INFERENCE tensors; only the gathered non-local branch routes them
            # through all_gather + cat before the product with logit_scale, the other two branches multiply them directly and
            # autograd refuses ("Inference tensors cannot be saved for backward"): the reference's stage 2 runs in branch "g" only.
            try:
                s2(x, ids, mask, labels, s, global_step=0)
                raise AssertionError(f"reference stage 2 was expected to raise in branch {ftag}")
            except RuntimeError as e:
                assert "Inference tensors" in str(e), e
            fx[f"s2_{ftag}_reference_raises"] = "RuntimeError: Inference tensors cannot be saved for backward"
            continue
        # ---- stage 2, every schedule branch (CLIP_stage2.py:141-149: < 5000, >= 5000, the bare-except None)
        for step in (0, 1000, 5000, 6000, None):
            rr = s2(x, ids, mask, labels, s, global_step=step)
            ro = ora(x, t1, t2, s, global_step=step)
            worst = max(worst, _eq(ro["loss"].detach(), rr["loss"].detach(), f"stage-2 loss {ftag} {step}"),
                        _eq(ro["logits"].detach(), rr["logits"].detach(), f"stage-2 logits {ftag} {step}"))
            fx[f"s2_{ftag}_loss_{step}"] = float(rr["loss"].detach())
            fx[f"s2_{ftag}_logits_{step}"] = rr["logits"].detach().numpy()
            if step == 1000:
                s2.zero_grad(), ora.zero_grad()
                rr["loss"].backward(), ro["loss"].backward()
                named_r = [(n, p) for n, p in s2.named_parameters() if p.requires_grad and "language" not in n]
                named_o = [(n, p) for n, p in ora.named_parameters() if p.requires_grad]
                worst = max(worst, _grads_equal(named_o, named_r, f"stage-2 CLIP step {ftag}"))
                fx.update(_grad_fixture(named_r, f"s2{ftag}grad"))
                fx[f"s2_{ftag}_loss_cl"], fx[f"s2_{ftag}_loss_relation"] = float(ro["loss_CL_stage2"]), float(ro["loss_relation"])
    dist.destroy_process_group()
    # ---- world size 2: two gloo processes run the reference; the oracle restates each rank's view from the global batch
    world, Bl = 2, 2
    sd_ = (111, 112, 113, 114)
    cases = {"s2g": (2, img, patch, L, Bl, 1000, True, False, sd_), "s1g": (1, img, patch, L, Bl, 1000, True, False, sd_),
             "s1l": (1, img, patch, L, Bl, 1000, True, True, sd_), "s1n": (1, img, patch, L, Bl, 1000, False, False, sd_)}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_clip_world_worker, args=(r, world, 29547, cases, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=600) for _ in range(world))
    [p.join() for p in procs]
    for tag, (stage, img_, patch_, L_, B_, step, gl, ll, _) in cases.items():
        s2, ora = _ref_clip_models(c1, c2, img_, patch_, L_, gl, ll, sd_[0])
        xg, sg = R.synthetic_volume((B_ * world, 1) + img_, sd_[1]), R.synthetic_slices(B_ * world, sd_[2])
        idg, mg = _tokens(B_ * world, sd_[3])
        with torch.no_grad():
            t1 = s2.stage1_pretrained_CLIP.encode_text(idg, mg)[0][:, 0].clone()
            t2 = s2.encode_text(idg, mg)[0][:, 0].clone()
        if stage == 1:
            ora = ora.stage1_pretrained_CLIP
            ora.requires_grad_(True)
            ora.local_loss, ora.gather_loss = ll, gl
        ora.zero_grad()
        total = 0
        for r in range(world):
            ro = ora(xg, t1, global_step=step, world=world, rank=r) if stage == 1 else ora(xg, t1, t2, sg, global_step=step, world=world, rank=r)
            worst = max(worst, _eq(ro["loss"].detach(), torch.tensor(got[r][tag]["loss"]), f"world-2 {tag} rank {r} loss"),
                        _eq(ro["logits"].detach(), torch.from_numpy(got[r][tag]["logits"]), f"world-2 {tag} rank {r} logits"))
            total = total + ro["loss"] / world
            fx[f"w2_{tag}_loss_r{r}"], fx[f"w2_{tag}_logits_r{r}"] = got[r][tag]["loss"], got[r][tag]["logits"]
        total.backward()                        # DDP: the mean over ranks of each rank's gradient (all_gather's backward sums the ranks' contributions)
        ngrad = 0
        for n, p in ora.named_parameters():
            if p.requires_grad and p.grad is not None and n in got[0][tag]["grads"]:
                avg = torch.from_numpy(np.asarray(sum(got[r][tag]["grads"][n] for r in range(world)) / world, dtype=np.float32))
                den = float(avg.double().norm())
                if den > 1e-9:
                    e = float((p.grad.double() - avg.double()).norm()) / den
                    assert e < 1e-4, (tag, n, e)      # a rank runs half the batch: other GEMM blocking than the oracle's global batch
                    worst = max(worst, e)
                    ngrad += 1
                    if n.endswith("mm_vision_proj.weight") or n.endswith("cls_token"):
                        a2 = avg.reshape(-1, avg.shape[-1])
                        fx[f"w2_{tag}_g/{n}"] = a2[sorted({0, a2.shape[0] // 2, a2.shape[0] - 1})].numpy()     # rows [0, mid, last]
        assert ngrad > 20, (tag, ngrad)
        fx.update({"w2_text1": t1.numpy(), "w2_text2": t2.numpy()})
    fx.update(w2_seeds=np.array([111, 112, 113, 114]), w2_batch_local=Bl, w2_step=1000)
    np.savez_compressed(os.path.join(OUT, "clip_ref.npz"), **fx)
    print(f"clip_ref ok: oracle == reference M3DCLIP_stage1 / M3DCLIP_stage2 in every branch (worst {worst:.1e}); "
          f"{len(fx['stage2_keys'])} state-dict keys recorded")


def clip_c4a():
    """BASELINE configs[3] at full geometry (256^3 / 16^3, L = 12), 2 volumes, global step 1000: loss, logits and the gradient of every
    trainable VISION tensor from the REFERENCE's own M3DCLIP_stage2.forward + autograd (round 5; rounds 2-4 wrote this file from the
    restatement).  The text features the reference's two (random-init, one-layer) BERT towers produced are stored: they are the
    inputs of the HIP step and of the oracle.  oracle/clip_cpu.py is asserted equal before anything is written."""
    import torch.distributed as dist
    c1, c2 = load_reference_clip()
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29543", rank=0, world_size=1)
    img, patch, B, step = (256, 256, 256), (16, 16, 16), 2, 1000
    s2, ora = _ref_clip_models(c1, c2, img, patch, 12, True, False, 91)
    for m in (s2, ora):                       # the committed fixture's convention: both scales log(1/0.07)
        with torch.no_grad():
            m.stage1_pretrained_CLIP.logit_scale.fill_(float(np.log(1 / 0.07)))
    x, s = R.synthetic_volume((B, 1) + img, 92), R.synthetic_slices(B, 93)
    ids, mask = _tokens(B, 94)
    with torch.no_grad():
        t1 = s2.stage1_pretrained_CLIP.encode_text(ids, mask)[0][:, 0].clone()
        t2 = s2.encode_text(ids, mask)[0][:, 0].clone()
    rr = s2(x, ids, mask, torch.arange(B), s, global_step=step)
    rr["loss"].backward()
    ro = ora(x, t1, t2, s, global_step=step)
    ro["loss"].backward()
    _eq(ro["loss"].detach(), rr["loss"].detach(), "configs[3] loss"), _eq(ro["logits"].detach(), rr["logits"].detach(), "configs[3] logits")
    named = [(n, p) for n, p in s2.named_parameters() if p.requires_grad and "language" not in n]
    worst = _grads_equal([(n, p) for n, p in ora.named_parameters() if p.requires_grad], named, "configs[3] step")
    fx = _grad_fixture(named, "grad")
    fx.update(img=np.array(img), patch=np.array(patch), batch=B, global_step=step, seed_w=91, seed_x=92, seed_s=93, seed_tok=94,
              text1=t1.numpy(), text2=t2.numpy(), loss=float(rr["loss"].detach()), loss_cl=float(ro["loss_CL_stage2"]),
              loss_relation=float(ro["loss_relation"]), logits=rr["logits"].detach().numpy(), source="reference M3DCLIP_stage2.forward + autograd")
    np.savez_compressed(os.path.join(OUT, "clip_c4a_256.npz"), **fx)
    dist.destroy_process_group()
    print(f"clip_c4a_256 ok (REFERENCE forward + autograd): loss {float(rr['loss']):.6f}; oracle gradients vs reference worst {worst:.1e}; "
          f"{len(named)} trainable vision tensors")


# --------------------------------------------------------------------------------------
# the LamedMetaForCausalLM mixin (lamed_arch.py:114-155) and the CT script's own arithmetic (round 5)
# --------------------------------------------------------------------------------------
@torch.no_grad()
def mixin_ref():
    """tests/golden/mixin_ref.npz: the REFERENCE's LamedMetaForCausalLM.encode_images / prepare_inputs_for_multimodal
    (lamed_arch.py:122-155) over the reference's build_vision_tower / build_mm_projector (multimodal_encoder/builder.py:4-11,
    multimodal_projector/builder.py:81-106), native 32x256x256 geometry, towers cut to 2 blocks.  lamed_arch.py imports the
    out-of-scope segmentation module (-> monai.networks.nets) at module scope: that one builder module gets an empty stand-in."""
    from oracle import clip_cpu as RC
    load_reference_clip()
    name = "LaMed.src.model.segmentation_module.builder"
    if name not in sys.modules:
        st = types.ModuleType(name)
        st.build_segmentation_module = lambda config, **k: None
        sys.modules[name] = st
    import LaMed.src.model.lamed_arch as la
    img, patch, L, B, S = (32, 256, 256), (4, 16, 16), 2, 1, 300

    class Cfg2:
        vision_tower, vision_select_layer, vision_select_feature, image_channel = "vit_stage2_dual_encoders", -1, "patch", 1
        remain_2d3d_ViT_type, image_size, patch_size = "dual_vits", img, patch
        mm_projector_type, mm_hidden_size, hidden_size = "VisualPacker_3d_phi_v3", 768, 3072
        proj_layer_type, proj_layer_num, proj_pooling_type, proj_pooling_size = "mlp", 2, "spatial", 2

    def build(tower_fn, proj_fn, parallel):
        tower = tower_fn(Cfg2()).eval()
        for t in (tower.vision_tower_stage1, tower.vision_tower_stage2):
            del t.blocks[L:]
        p1, p2 = proj_fn(Cfg2()).eval(), proj_fn(Cfg2()).eval()
        for m_, seed in ((tower, 201), (p1, 202), (p2, 203)):
            m_.load_state_dict(R.seeded_state_dict(m_, seed))
        emb = torch.nn.Embedding(64, 3072)
        emb.load_state_dict(R.seeded_state_dict(emb, 204))

        class Model:
            mm_projector = p1
            embed_tokens = emb

            def get_vision_tower(self):
                return tower
        model = Model()
        if parallel:
            model.mm_projector2 = p2
        return model, tower, p1, p2

    x, s = R.synthetic_volume((B, 1) + img, 205), R.synthetic_slices(B, 206)
    ids = torch.from_numpy(np.random.default_rng(207).integers(0, 64, size=(B, S)))
    fx = dict(img=np.array(img), patch=np.array(patch), layers=L, batch=B, seq=S, seed_tower=201, seed_p1=202, seed_p2=203, seed_emb=204,
              seed_x=205, seed_s=206, seed_ids=207)
    rows = [0, 1, 127, 128, 255]
    for parallel, tag in ((True, "par"), (False, "shared")):
        model, _, _, _ = build(la.build_vision_tower, la.build_mm_projector, parallel)

        class LM(la.LamedMetaForCausalLM):
            def get_model(self):
                return model
        lm = LM()
        feats = lm.encode_images(x, None, s)                                                     # :122-141
        r = lm.prepare_inputs_for_multimodal(ids, "pos", "mask", "pkv", "labels", x, s)          # :143-155
        assert r[0] is None and r[1:4] == ("pos", "mask", "pkv") and r[5] == "labels"
        emb = r[4]
        # the oracle's restatement of the same two steps
        om, ot, op1, op2 = build(R.ViT3DTower_dual_encoders, lambda c: R.VisualPacker_3d_phi_v3(c.image_size, c.patch_size, c.mm_hidden_size, c.hidden_size,
                                                                                             c.proj_layer_type, c.proj_layer_num), parallel)
        of = R.encode_images(ot, op1, op2 if parallel else None, x, s)
        oe = RC.splice_image_features(om.embed_tokens(ids), of)
        _eq(of, feats, f"mixin encode_images ({tag})"), _eq(oe, emb, f"mixin splice ({tag})")
        fx.update({f"{tag}_feat_rows": _sel(feats, rows), f"{tag}_feat_norms": feats.norm(dim=-1).numpy(),
                   f"{tag}_emb_rows": _sel(emb, [0, 1, 2, 256, 257, S - 1]), f"{tag}_emb_norms": emb.norm(dim=-1).numpy()})
        # pass-through cases (:148-149)
        one = torch.zeros(B, 1, dtype=torch.long)
        assert lm.prepare_inputs_for_multimodal(one, 1, 2, 3, 4, x, s) == (one, 1, 2, 3, None, 4)
        assert lm.prepare_inputs_for_multimodal(ids, 1, 2, 3, 4, None, s)[0] is ids
    fx.update(feat_rows_idx=np.array(rows), emb_rows_idx=np.array([0, 1, 2, 256, 257, S - 1]))
    np.savez_compressed(os.path.join(OUT, "mixin_ref.npz"), **fx)
    print("mixin_ref ok: oracle encode_images / splice == reference LamedMetaForCausalLM (parallel and shared projector)")


def prep_ref():
    """tests/golden/prep_ref.npz: the CT script's own `resize_array` and `nii_img_to_tensor`
    (Data/data_processing/CT-RATE/CT-RATE_nii_to_3D_volume_npy_file.py:24-117) run here on a synthetic scan.  The script imports
    monai.transforms / nibabel at module scope (absent: empty stand-ins; its module-level mtf.Compose is never called here because
    `transform` is an ARGUMENT of nii_img_to_tensor: the oracle's crop + resize is passed in) and moves tensors through 'cuda'
    (absent here: Tensor.to is wrapped for the duration of the call so that 'cuda' means this CPU; the arithmetic is device-neutral)."""
    from oracle import prep_cpu as P
    import pandas as pd
    for n in ("monai.transforms", "nibabel"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules.setdefault("monai", types.ModuleType("monai"))
    mtf = sys.modules["monai.transforms"]
    mtf.Compose = lambda ts: ts
    mtf.CropForeground = lambda *a, **k: None
    mtf.Resize = lambda *a, **k: None
    path = "/root/reference/Data/data_processing/CT-RATE/CT-RATE_nii_to_3D_volume_npy_file.py"
    ct = _load(path, "_ref_ct_script")
    g = np.random.default_rng(301)
    # (i) resize_array alone
    arr = torch.from_numpy(g.standard_normal((1, 1, 11, 20, 24))).double()
    a_ref, a_ora = ct.resize_array(arr, (3.0, 0.9, 0.9), (1.5, 0.75, 0.75)), P.resize_array(arr, (3.0, 0.9, 0.9), (1.5, 0.75, 0.75))
    assert a_ref.shape == a_ora.shape == (1, 1, 22, 24, 28) and np.array_equal(a_ref, a_ora)
    # (ii) the whole function on a synthetic scan (H, W, D) = (40, 48, 14): a body of tissue values inside air
    raw = np.full((40, 48, 14), -1200.0)
    raw[6:33, 9:41, 2:12] = g.uniform(-900, 700, size=(27, 32, 10))
    slope, intercept, zsp, xysp = 1.0, -24.0, 3.0, 0.9

    class _Nii:
        def get_fdata(self):
            return raw.copy()
    sys.modules["nibabel"].load = lambda p: _Nii()
    ct.nib.load = sys.modules["nibabel"].load
    meta = pd.DataFrame([dict(VolumeName="scan.nii.gz", RescaleSlope=slope, RescaleIntercept=intercept, XYSpacing=f"[{xysp}, {xysp}]", ZSpacing=zsp)])
    out_size = (8, 32, 32)
    transform = lambda t: P.monai_resize(P.crop_foreground(t), out_size)
    real_to, real_exists = torch.Tensor.to, os.path.exists

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(v, str) and v.startswith("cuda")) else v for v in a)
        return real_to(self, *a, **k)
    try:
        torch.Tensor.to = to_cpu
        ct.os.path.exists = lambda p: True
        y_ref = ct.nii_img_to_tensor("/nowhere/scan.nii.gz", transform, meta)
    finally:
        torch.Tensor.to = real_to
        ct.os.path.exists = real_exists
    assert y_ref is not None, "the reference function reported an error (printed above)"
    y_ora = P.nii_img_to_tensor(raw, slope, intercept, zsp, xysp, out_size)
    assert y_ref.shape == y_ora.shape == (1,) + out_size and torch.equal(y_ref, y_ora), float((y_ref - y_ora).abs().max())
    np.savez_compressed(os.path.join(OUT, "prep_ref.npz"), seed=301, raw_shape=np.array(raw.shape), body=np.array([6, 33, 9, 41, 2, 12]),
                        slope=slope, intercept=intercept, z_spacing=zsp, xy_spacing=xysp, out_size=np.array(out_size), out=y_ref.numpy(),
                        resized=a_ref[0, 0, ::7, ::5, ::9])
    print("prep_ref ok: oracle resize_array / nii_img_to_tensor == the reference script's (bit-equal)")


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
    ap.add_argument("--only", default=None, choices=["grads", "clip", "mixin", "prep"], help="regenerate only this group of fixtures")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present (this script only runs in the build container)")
    print(f"[make_goldens] rearrange pattern witnessed by {check_rearrange_witness()} reference comments", flush=True)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    vit, spp = load_reference()
    if a.only == "grads":
        grads(vit, spp)
        return
    if a.only == "mixin":
        mixin_ref()
        return
    if a.only == "prep":
        prep_ref()
        return
    if a.only == "clip":
        clip_ref()
        if not a.skip_big:
            clip_c4a()
        return
    tiny_encoders(vit)
    packer(spp)
    plumbing(vit)
    if not a.skip_big:
        big(vit, spp)
        grads(vit, spp)
    clip_ref()
    mixin_ref()
    prep_ref()
    if not a.skip_big:
        clip_c4a()


if __name__ == "__main__":
    main()
