"""GPU parity against fixtures the REFERENCE's own wrapper classes wrote (round 5, oracle/make_goldens.py --only clip / mixin):
M3DCLIP_stage1.forward / M3DCLIP_stage2.forward (CLIP_stage1.py:111-155, CLIP_stage2.py:119-190) -> tests/golden/clip_ref.npz and
LamedMetaForCausalLM.encode_images / prepare_inputs_for_multimodal (lamed_arch.py:122-155) -> tests/golden/mixin_ref.npz; and N3:
a checkpoint with the reference's real state_dict() key list handed over positionally (train/train_VLM.py:477-503) into HIP modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _fixture_grad_errors(named, g, tag, with_global=False):
    """(worst per-tensor norm error, worst value error over the stored rows[, global error over every stored value]) against a
    _grad_fixture group; tensors whose reference norm is below 1e-3 of the largest are left out of the per-tensor figures."""
    names = [str(n) for n in g[f"{tag}_names"]]
    norms = dict(zip(names, g[f"{tag}_norms"]))
    wn, wv = ("", 0.0), ("", 0.0)
    big = max(norms.values())
    num = den = 0.0
    for n, p in named:
        key = f"{tag}_g/{n}"
        if key in g.files:
            want = torch.from_numpy(g[key]).double()
            got = (p.grad.reshape(-1, p.grad.shape[-1])[g[f"{tag}_r/{n}"]] if f"{tag}_r/{n}" in g.files else p.grad).reshape(want.shape).double().cpu()
            num += float((got - want).pow(2).sum())
            den += float(want.pow(2).sum())
    if with_global:
        return _fixture_grad_errors(named, g, tag) + ((num / den) ** 0.5,)
    for n, p in named:
        if norms[n] < 1e-3 * big:
            continue
        gn = float(p.grad.double().norm())
        wn = max(wn, (n, abs(gn - norms[n]) / norms[n]), key=lambda t: t[1])
        key = f"{tag}_g/{n}"
        if key in g.files:
            want = torch.from_numpy(g[key])
            got = p.grad.reshape(-1, p.grad.shape[-1])[g[f"{tag}_r/{n}"]] if f"{tag}_r/{n}" in g.files else p.grad
            wv = max(wv, (n, rel(got.reshape(want.shape), want)), key=lambda t: t[1])
    return wn, wv


def _clip_models(dev, g, gather_loss=True, local_loss=False):
    from hsenet_amd.clip import M3DCLIPVision_stage2
    from oracle import ref_cpu as R
    img, patch, L = tuple(int(v) for v in g["img"]), tuple(int(v) for v in g["patch"]), int(g["layers"])
    m = M3DCLIPVision_stage2(img, patch, num_layers=L, gather_loss=gather_loss, local_loss=local_loss)
    sd = R.seeded_state_dict(m, int(g["seed_w"]))
    sd["logit_scale"] = torch.tensor(float(g["scale_outer"]))
    sd["stage1_pretrained_CLIP.logit_scale"] = torch.tensor(float(g["scale_inner"]))
    m.load_state_dict(sd)
    return m.to(dev).eval(), img


@pytest.mark.parametrize("prec,tol_f,tol_g", [("f32", 2e-5, 3e-4), ("bf16", 2e-2, None)])
def test_clip_wrappers_vs_reference_run(dev, golden, prec, tol_f, tol_g):
    """the module API's stage-1 step in all three branches (gathered / local_loss / no gather) and the stage-2 step over the
    relation schedule's branches, against the REFERENCE classes' own losses, logits and vision gradients."""
    import hsenet_amd as H
    from oracle import ref_cpu as R
    g = golden("clip_ref.npz")
    B = int(g["batch"])
    t1, t2 = torch.from_numpy(g["text1"]).to(dev), torch.from_numpy(g["text2"]).to(dev)
    H.set_precision(prec)
    try:
        for gl, ll, tag in ((True, False, "g"), (True, True, "l"), (False, False, "n")):
            m, img = _clip_models(dev, g, gl, ll)
            x, s = R.synthetic_volume((B, 1) + img, int(g["seed_x"])).to(dev), R.synthetic_slices(B, int(g["seed_s"])).to(dev)
            s1 = m.stage1_pretrained_CLIP
            s1.local_loss, s1.gather_loss = ll, gl
            s1.requires_grad_(True)
            r = s1(x, t1, global_step=500)
            r["loss"].backward()
            torch.cuda.synchronize()
            assert abs(float(r["loss"]) - float(g[f"s1_{tag}_loss"])) < tol_f * 5, (tag, float(r["loss"]), float(g[f"s1_{tag}_loss"]))
            assert rel(r["logits"].detach(), g[f"s1_{tag}_logits"]) < tol_f
            wn, wv, gl_ = _fixture_grad_errors(list(s1.named_parameters()), g, f"s1{tag}grad", True)
            print(f"stage-1 CLIP step [{prec}, branch {tag}] vs the reference run: loss {float(r['loss']):.6f} / {float(g[f's1_{tag}_loss']):.6f}; "
                  f"gradients global {gl_:.1e}, worst norm {wn[0]} {wn[1]:.1e}, worst value {wv[0]} {wv[1]:.1e}")
            assert (wn[1] < tol_g and wv[1] < tol_g) if tol_g else (gl_ < 6e-2 and wn[1] < 0.4 and wv[1] < 0.4)
            s1.requires_grad_(False)
            if tag != "g":
                continue                                   # the reference's stage 2 raises in these branches (fixture records it)
            for step in (0, 1000, 5000, 6000, None):
                m.zero_grad(set_to_none=True)
                r = m(x, t1, t2, s, global_step=step)
                assert abs(float(r["loss"]) - float(g[f"s2_g_loss_{step}"])) < tol_f * 5, step
                assert rel(r["logits"].detach(), g[f"s2_g_logits_{step}"]) < tol_f
                if step == 1000:
                    r["loss"].backward()
                    torch.cuda.synchronize()
                    wn, wv, gl_ = _fixture_grad_errors([(n, p) for n, p in m.named_parameters() if p.requires_grad], g, "s2ggrad", True)
                    print(f"stage-2 CLIP step [{prec}] vs the reference run: gradients global {gl_:.1e}, worst norm {wn[0]} {wn[1]:.1e}, worst value {wv[0]} {wv[1]:.1e}")
                    assert (wn[1] < tol_g and wv[1] < tol_g) if tol_g else (gl_ < 6e-2 and wn[1] < 0.4 and wv[1] < 0.4)
    finally:
        H.set_precision("bf16")


@pytest.mark.parametrize("precision,tol_loss,tol_glob,tol_each", [("bf16", 5e-3, 6e-2, 0.4), ("f16", 5e-4, 1e-2, 6e-2)])
def test_clip_engines_vs_reference_run(dev, golden, precision, tol_loss, tol_glob, tol_each):
    """the fused engines (the bench path of configs[3]) on the same fixture: ClipStage1Engine and ClipStage2Engine at bf16 and -- round 5 --
    at fp16 operands (dL/dlogits carries the engine's loss scale, divided out here before the comparison and inside AdamW in a step)."""
    from hsenet_amd.clip_step import ClipStage1Engine, ClipStage2Engine
    from oracle import ref_cpu as R
    g = golden("clip_ref.npz")
    B = int(g["batch"])
    t1, t2 = torch.from_numpy(g["text1"]).to(dev), torch.from_numpy(g["text2"]).to(dev)
    m, img = _clip_models(dev, g)
    x, s = R.synthetic_volume((B, 1) + img, int(g["seed_x"])).to(dev), R.synthetic_slices(B, int(g["seed_s"])).to(dev)
    eng = ClipStage2Engine(m, B, precision=precision)
    assert eng.loss_scale == (1.0 if precision == "bf16" else 2.0 ** 10) and eng.patches.dtype == (torch.bfloat16 if precision == "bf16" else torch.float16)
    eng.global_step = 1000
    eng.forward_loss(x, s, t1, t2)
    eng.backward()
    eng.grad.div_(eng.loss_scale)
    torch.cuda.synchronize()
    assert abs(eng.loss_value() - float(g["s2_g_loss_1000"])) < tol_loss * float(g["s2_g_loss_1000"])
    assert rel((eng.logits2 + eng.logits2.t()) / 2, g["s2_g_logits_1000"]) < 4 * tol_loss
    wn, wv, gl_ = _fixture_grad_errors([(n, p) for n, p in m.named_parameters() if p.requires_grad], g, "s2ggrad", True)
    print(f"ClipStage2Engine [{precision}] vs the reference run: loss {eng.loss_value():.6f} / {float(g['s2_g_loss_1000']):.6f}; gradients global {gl_:.1e}, "
          f"worst norm {wn[0]} {wn[1]:.1e}, worst value {wv[0]} {wv[1]:.1e}")
    assert gl_ < tol_glob and wn[1] < tol_each and wv[1] < tol_each and bool(torch.isfinite(eng.grad).all())
    del eng
    m1, _ = _clip_models(dev, g)
    s1 = m1.stage1_pretrained_CLIP
    s1.requires_grad_(True)
    e1 = ClipStage1Engine(s1, B, precision=precision)
    e1.forward_loss(x, t1)
    e1.backward()
    e1.grad.div_(e1.loss_scale)
    torch.cuda.synchronize()
    assert abs(e1.loss_value() - float(g["s1_g_loss"])) < tol_loss * float(g["s1_g_loss"])
    wn, wv, gl_ = _fixture_grad_errors(list(s1.named_parameters()), g, "s1ggrad", True)
    print(f"ClipStage1Engine [{precision}] vs the reference run: loss {e1.loss_value():.6f} / {float(g['s1_g_loss']):.6f}; gradients global {gl_:.1e}, "
          f"worst norm {wn[0]} {wn[1]:.1e}, worst value {wv[0]} {wv[1]:.1e}")
    assert gl_ < tol_glob and wn[1] < tol_each and wv[1] < tol_each
    if precision == "f16":          # a whole fp16 step trains: the optimizer takes the scale out again
        e1.grad.mul_(e1.loss_scale)
        before = e1.flat.clone()
        e1.optimizer_step()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(e1.flat).all()) and float((e1.flat - before).abs().max()) > 0


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16", 2e-2), ("f16", 3e-3)])
def test_mixin_vs_reference_run(dev, golden, prec, tol):
    """hsenet_amd.LamedMetaForCausalLM over build_vision_tower / build_mm_projector against the reference mixin's own outputs
    (native geometry, towers of 2 blocks): parallel projectors and the shared-projector fallback, then the splice."""
    import hsenet_amd as H
    from oracle import ref_cpu as R
    g = golden("mixin_ref.npz")
    img, patch, L, B, S = tuple(int(v) for v in g["img"]), tuple(int(v) for v in g["patch"]), int(g["layers"]), int(g["batch"]), int(g["seq"])

    class Cfg:
        vision_tower, vision_select_layer, vision_select_feature, image_channel = "vit_stage2_dual_encoders", -1, "patch", 1
        remain_2d3d_ViT_type, image_size, patch_size = "dual_vits", img, patch
        mm_projector_type, mm_hidden_size, hidden_size = "VisualPacker_3d_phi_v3", 768, 3072
        proj_layer_type, proj_layer_num, proj_pooling_type, proj_pooling_size = "mlp", 2, "spatial", 2
    tower = H.build_vision_tower(Cfg()).eval()
    for t in (tower.vision_tower_stage1, tower.vision_tower_stage2):
        del t.blocks[L:]
    p1, p2 = H.build_mm_projector(Cfg()).eval(), H.build_mm_projector(Cfg()).eval()
    emb = torch.nn.Embedding(64, 3072)
    for m_, key in ((tower, "seed_tower"), (p1, "seed_p1"), (p2, "seed_p2"), (emb, "seed_emb")):
        m_.load_state_dict(R.seeded_state_dict(m_, int(g[key])))
        m_.to(dev)
    x, s = R.synthetic_volume((B, 1) + img, int(g["seed_x"])).to(dev), R.synthetic_slices(B, int(g["seed_s"])).to(dev)
    ids = torch.from_numpy(np.random.default_rng(int(g["seed_ids"])).integers(0, 64, size=(B, S))).to(dev)

    def lm_for(parallel):
        class Model:
            mm_projector, embed_tokens = p1, emb

            def get_vision_tower(self):
                return tower
        model = Model()
        if parallel:
            model.mm_projector2 = p2

        class LM(H.LamedMetaForCausalLM):
            def get_model(self):
                return model
        return LM()
    H.set_precision(prec)
    try:
        with torch.no_grad():
            for tag, parallel in (("par", True), ("shared", False)):
                lm = lm_for(parallel)
                f = lm.encode_images(x, None, s)
                r = lm.prepare_inputs_for_multimodal(ids, "pos", "mask", "pkv", "labels", x, s)
                assert r[0] is None and r[1:4] == ("pos", "mask", "pkv") and r[5] == "labels"
                e = r[4].float()
                ef, nf = rel(f[:, g["feat_rows_idx"]], g[f"{tag}_feat_rows"]), rel(f.norm(dim=-1), g[f"{tag}_feat_norms"])
                ee, ne = rel(e[:, g["emb_rows_idx"]], g[f"{tag}_emb_rows"]), rel(e.norm(dim=-1), g[f"{tag}_emb_norms"])
                print(f"mixin [{prec}, {tag}] vs the reference mixin: features {ef:.1e} (norms {nf:.1e}), spliced embeddings {ee:.1e} (norms {ne:.1e})")
                assert max(ef, nf, ee, ne) < tol
    finally:
        H.set_precision("bf16")


def test_reference_layout_checkpoint_into_hip_modules(dev, golden, tmp_path):
    """N3 on the device: a CLIP checkpoint with the REFERENCE's own M3DCLIP_stage2.state_dict() key list and shapes
    (recorded from the class, incl. the BERT tower's keys and buffers) is saved in the reference's layout (train_CLIP_stage1.py:
    198-216), read back, filtered (train_VLM.py:478) and copied BY POSITION (:479-503) into a dual tower of HIP modules on the GPU;
    its forward must equal the oracle tower loaded from the same file by the same rule.  Then the projector-only checkpoint of
    lamed_trainer.py:20-23 round-trips through HIP packers."""
    import ast
    import hsenet_amd as H
    from hsenet_amd import checkpoint as C
    from oracle import ref_cpu as R
    g = golden("clip_ref.npz")
    keys, shapes = [str(k) for k in g["stage2_keys"]], [ast.literal_eval(str(v)) for v in g["stage2_shapes"]]
    img, patch, L = tuple(int(v) for v in g["img"]), tuple(int(v) for v in g["patch"]), int(g["layers"])
    rng = np.random.default_rng(4242)
    sd = {}
    for k, shp in zip(keys, shapes):
        if k.endswith("position_ids") or k.endswith("token_type_ids"):
            sd[k] = torch.zeros(shp, dtype=torch.long)
            continue
        a = rng.standard_normal(shp).astype(np.float32) * 0.02
        if (".norm" in k or "LayerNorm" in k) and k.endswith("weight"):
            a = 1.0 + 2.5 * a
        sd[k] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    ck = str(tmp_path / "saved_1000")
    C.save_clip_checkpoint(ck, sd)
    loaded = C.load_clip_state_dict(ck)
    assert list(loaded.keys()) == keys
    vis = C.filter_visual_checkpoint(loaded)

    class Cfg:
        vision_tower, vision_select_layer, vision_select_feature, image_channel = "vit_stage2_dual_encoders", -1, "patch", 1
        remain_2d3d_ViT_type, image_size, patch_size = "dual_vits", img, patch
    towers = []
    for make, where in ((H.build_vision_tower, dev), (R.ViT3DTower_dual_encoders, "cpu")):
        t = make(Cfg()).eval()
        for e in (t.vision_tower_stage1, t.vision_tower_stage2):
            del e.blocks[L:]
        t = t.to(where)
        n1, n2, skipped = C.load_towers_positional(t, vis)              # the HIP tower receives its tensors ON THE DEVICE
        assert (n1, n2, skipped) == (1 + 3 + 11 * L + 2, 1 + 3 + 11 * L + 2 + 12, [])
        towers.append(t)
    hip, ora = towers
    assert torch.equal(hip.vision_tower_stage2.slice_guided_attention.Wq.weight.cpu(), loaded["vision_encoder.slice_guided_attention.Wq.weight"])
    assert torch.equal(hip.vision_tower_stage1.blocks[1].attn.qkv.weight.cpu(), loaded["stage1_pretrained_CLIP.vision_encoder.blocks.1.attn.qkv.weight"])
    x, s = R.synthetic_volume((2, 1) + img, 77), R.synthetic_slices(2, 78)
    with torch.no_grad():
        o1, o2 = ora(x, s)
        H.set_precision("f32")
        try:
            h1, h2 = hip(x.to(dev), s.to(dev))
        finally:
            H.set_precision("bf16")
        b1, b2 = hip(x.to(dev), s.to(dev))
    e32, e16 = max(rel(h1, o1), rel(h2, o2)), max(rel(b1, o1), rel(b2, o2))
    print(f"reference-layout checkpoint -> HIP dual tower on the device: forward vs the oracle loaded from the same file f32 {e32:.1e}, bf16 {e16:.1e}")
    assert e32 < 1e-3 and e16 < 2e-2
    # projector / LoRA checkpoint (lamed_trainer.py:20-23, train_VLM.py:505-510)
    pa = (img, patch, 768, 256, "mlp", 2)
    pk = dict(grid=(2, 4, 4), kernel_size=(1, 2, 2))

    class MLLM(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            self.vision_tower = torch.nn.Linear(2, 2)
            self.mm_projector, self.mm_projector2 = H.VisualPacker_3d_phi_v3(*pa, **pk), H.VisualPacker_3d_phi_v3(*pa, **pk)
            self.lora_A = torch.nn.Parameter(torch.zeros(3, 3))
            for i, m_ in enumerate((self.mm_projector, self.mm_projector2)):
                m_.load_state_dict(R.seeded_state_dict(m_, seed + i))
            with torch.no_grad():
                self.lora_A.fill_(float(seed))
    src, dst = MLLM(10).to(dev).eval(), MLLM(20).to(dev).eval()     # eval(): the packers' Dropout(0.1) sites draw a fresh mask per call in train()
    f = C.save_projector_and_lora(str(tmp_path / "mllm"), src)
    saved = torch.load(f, map_location="cpu", weights_only=True)
    assert all(("mm_projector" in k) or ("lora" in k) for k in saved) and not any("vision_tower" in k for k in saved)
    res = C.load_projector_and_lora(dst, f)
    assert not res.unexpected_keys and all("vision_tower" in k for k in res.missing_keys)
    feats = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 32, 768), dtype=np.float32)).to(dev)
    with torch.no_grad():
        assert torch.equal(src.mm_projector2(feats), dst.mm_projector2(feats)) and float(dst.lora_A[0, 0]) == 10.0


# ---- world size 2: the product over two ranks (gloo, both on cuda:0) against the REFERENCE's own two-process run -------------------------
def _w2_worker(rank, world, port, out, golden_path):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hsenet_amd as H
        from oracle import ref_cpu as R
        g = np.load(golden_path, allow_pickle=False)
        dev = torch.device("cuda:0")
        Bl, step = int(g["w2_batch_local"]), int(g["w2_step"])
        sw, sx, ss, _ = (int(v) for v in g["w2_seeds"])
        t1, t2 = torch.from_numpy(g["w2_text1"]).to(dev), torch.from_numpy(g["w2_text2"]).to(dev)
        sl = slice(rank * Bl, (rank + 1) * Bl)
        res = {}
        H.set_precision("f32")
        for tag, stage, gl, ll in (("s2g", 2, True, False), ("s1g", 1, True, False), ("s1l", 1, True, True), ("s1n", 1, False, False)):
            m, img = _clip_models(dev, dict(img=g["img"], patch=g["patch"], layers=g["layers"], seed_w=sw, scale_outer=g["scale_outer"],
                                           scale_inner=g["scale_inner"]), gl, ll)
            x, s = R.synthetic_volume((Bl * world, 1) + img, sx).to(dev), R.synthetic_slices(Bl * world, ss).to(dev)
            if stage == 1:
                m = m.stage1_pretrained_CLIP
                m.local_loss, m.gather_loss = ll, gl
                m.requires_grad_(True)
                r = m(x[sl].contiguous(), t1[sl].contiguous(), global_step=step)
            else:
                r = m(x[sl].contiguous(), t1[sl].contiguous(), t2[sl].contiguous(), s[sl].contiguous(), global_step=step)
            r["loss"].backward()
            torch.cuda.synchronize()
            grads = {}
            for n, p in m.named_parameters():
                if p.grad is not None and (n.endswith("mm_vision_proj.weight") or n.endswith("cls_token")):
                    gsum = p.grad.detach().clone()
                    dist.all_reduce(gsum)                                     # what DDP does (then / world)
                    a2 = (gsum / world).reshape(-1, gsum.shape[-1])
                    grads[n] = a2[sorted({0, a2.shape[0] // 2, a2.shape[0] - 1})].cpu()
            res[tag] = dict(loss=float(r["loss"].detach()), logits=r["logits"].detach().cpu(), grads=grads)
        H.set_precision("bf16")
        torch.save(res, out + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_clip_wrappers_world2_vs_reference_two_process_run(dev, golden, tmp_path):
    """two ranks of the PRODUCT (module API, f32 mode; gloo between two processes sharing cuda:0) against what two gloo processes of the
    REFERENCE computed: each rank's loss and logits for stage 2 (gathered) and stage 1 in all three branches (gathered, local_loss with
    rank-offset labels, no gather), and the DDP-mean gradient of mm_vision_proj.weight / cls_token -- dp.gather_features' forward AND its
    backward (all-reduce of the gathered features' gradient) against torch.distributed.nn.all_gather's."""
    import os
    import socket
    import torch.multiprocessing as mp
    from conftest import GOLDEN
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "w2")
    mp.spawn(_w2_worker, args=(2, port, out, os.path.join(GOLDEN, "clip_ref.npz")), nprocs=2, join=True)
    g = golden("clip_ref.npz")
    worst = 0.0
    for rank in range(2):
        res = torch.load(out + f".{rank}")
        for tag in ("s2g", "s1g", "s1l", "s1n"):
            e_l = abs(res[tag]["loss"] - float(g[f"w2_{tag}_loss_r{rank}"]))
            e_g = rel(res[tag]["logits"], g[f"w2_{tag}_logits_r{rank}"])
            assert e_l < 1e-5 and e_g < 2e-5, (tag, rank, e_l, e_g)
            for n, got in res[tag]["grads"].items():
                key = f"w2_{tag}_g/{n}"
                if key in g.files:
                    e = rel(got, g[key])
                    worst = max(worst, e)
                    assert e < 3e-4, (tag, rank, n, e)
    print(f"product at world size 2 vs the reference's two-process run: losses / logits equal to 2e-5, DDP-mean gradients within {worst:.1e}")
