"""N1 and BASELINE configs[3] on the GPU: the CLIP loss head (both contrastive branches, relation MSE + schedule, the embedding
splice) and the fused stage-2 CLIP step engine (hsenet_amd/clip_step.py) against the CPU oracle (oracle/clip_cpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IMG, PATCH, LAYERS = (8, 64, 64), (4, 16, 16), 2
SCALE = float(np.log(1 / 0.07))


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _feats(n, seed, d=768):
    from oracle import clip_cpu as RC
    return RC.synthetic_text_features(n, seed), RC.synthetic_text_features(n, seed + 1)


@pytest.mark.parametrize("n", [1, 3, 8, 80])
@pytest.mark.parametrize("local_loss", [False, True])
def test_contrastive_and_relation_vs_oracle(dev, n, local_loss):
    """loss = CE(stage-2) + w * relation MSE against detached stage-1 logits; gradients to both feature sets and logit_scale."""
    from hsenet_amd import clip
    from oracle import clip_cpu as RC
    i1, t1 = _feats(n, 700 + n)
    i2, t2 = _feats(n, 710 + n)
    w = RC.relation_loss_weight(1250)
    assert abs(w - 0.075) < 1e-12 and RC.relation_loss_weight(5000) == 0.0 and RC.relation_loss_weight(None) == 0.0
    assert clip.relation_loss_weight(1250) == w and clip.relation_loss_weight(7000) == 0.0 and clip.relation_loss_weight(None) == 0.0
    # oracle
    ir, tr, sr = i2.clone().requires_grad_(), t2.clone().requires_grad_(), torch.tensor(SCALE, requires_grad=True)
    _, a1, b1 = RC.image_text_contrastive_learning(i1, t1, sr.detach(), local_loss)
    lr_, a2, b2 = RC.image_text_contrastive_learning(ir, tr, sr, local_loss)
    rr = RC.image_text_relation_regulation(a1.detach(), b1.detach(), a2, b2)
    (lr_ + w * rr).backward()
    # HIP
    im, tm = i2.to(dev).requires_grad_(), t2.to(dev).requires_grad_()
    sm = torch.tensor(SCALE, device=dev, requires_grad=True)
    with torch.no_grad():
        _, c1, d1 = clip.image_text_contrastive_learning(i1.to(dev), t1.to(dev), sm, local_loss=local_loss)
    lm, c2, d2 = clip.image_text_contrastive_learning(im, tm, sm, local_loss=local_loss)
    rm = clip.image_text_relation_regulation(c1, d1, c2, d2)
    (lm + w * rm).backward()
    assert rel(c2, a2.detach()) < 1e-5 and rel(d2, b2.detach()) < 1e-5
    assert abs(float(lm) - float(lr_)) <= 1e-5 * abs(float(lr_)) + 1e-6 and abs(float(rm) - float(rr)) <= 1e-4 * abs(float(rr)) + 1e-6
    assert float((im.grad.cpu() - ir.grad).abs().max()) <= 1e-4 * float(ir.grad.abs().max()) + 1e-6
    assert float((tm.grad.cpu() - tr.grad).abs().max()) <= 1e-4 * float(tr.grad.abs().max()) + 1e-6
    assert abs(float(sm.grad) - float(sr.grad)) < 1e-4 * abs(float(sr.grad)) + 1e-6
    with pytest.raises(ValueError):
        clip.image_text_contrastive_learning(im, tm, sm, labels=torch.arange(n))


def test_splice_vs_oracle(dev):
    """prepare_inputs_for_multimodal's splice (lamed_arch.py:151-154): 256 image tokens into positions 1..256."""
    from hsenet_amd import clip
    from oracle import clip_cpu as RC
    g = np.random.default_rng(730)
    for S, n in ((300, 256), (257, 256), (200, 256), (12, 4)):
        emb = torch.from_numpy(g.standard_normal((2, S, 64), dtype=np.float32))
        feat = torch.from_numpy(g.standard_normal((2, n, 64), dtype=np.float32))
        er, fr = emb.clone().requires_grad_(), feat.clone().requires_grad_()
        outr = RC.splice_image_features(er, fr)
        wgt = torch.from_numpy(g.standard_normal(tuple(outr.shape), dtype=np.float32))
        (outr * wgt).sum().backward()
        em, fm = emb.to(dev).requires_grad_(), feat.to(dev).requires_grad_()
        out = clip.splice_image_features(em, fm)
        assert tuple(out.shape) == tuple(outr.shape)
        (out * wgt.to(dev)).sum().backward()
        assert torch.equal(out.detach().cpu(), outr.detach())                      # pure data movement: bit-exact
        assert torch.equal(em.grad.cpu(), er.grad) and torch.equal(fm.grad.cpu(), fr.grad)


def _models(dev, seed, train=False):
    """oracle + HIP stage-2 CLIP (vision halves) with identical seeded weights and logit_scale = log(1/0.07)."""
    from hsenet_amd.clip import M3DCLIPVision_stage2
    from oracle import clip_cpu as RC, ref_cpu as R
    ref = RC.M3DCLIPVision_stage2(IMG, PATCH, LAYERS).eval()
    sd = R.seeded_state_dict(ref, seed)
    sd["logit_scale"] = torch.tensor(SCALE)
    sd["stage1_pretrained_CLIP.logit_scale"] = torch.tensor(SCALE)
    ref.load_state_dict(sd)
    m = M3DCLIPVision_stage2(IMG, PATCH, num_layers=LAYERS)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(sd)
    m = m.to(dev)
    m.train(train)
    return ref, m


def _grad_check(named_mine, named_ref, what):
    want = {n: p.grad.double() for n, p in named_ref if p.grad is not None}
    mine = {n: p.grad.double().cpu() for n, p in named_mine if n in want}
    assert sorted(mine) == sorted(want)
    num = sum(float((mine[n] - want[n]).pow(2).sum()) for n in want) ** 0.5
    den = sum(float(want[n].pow(2).sum()) for n in want) ** 0.5
    big = max(float(want[n].norm()) for n in want)
    per = {n: float((mine[n] - want[n]).norm() / want[n].norm()) for n in want if float(want[n].norm()) > 2e-2 * big}
    worst = max(per, key=per.get)
    print(f"{what}: global grad err {num / den:.2e}; worst of {len(per)} major tensors {worst} {per[worst]:.2e}")
    # the cls-only contrastive loss: per-sample gradients largely cancel over the batch, so the bf16 policy itself puts the fp32
    # oracle and its own bf16-autocast emulation 2.6e-2 apart globally (tests/test_gpu_parity.py::test_config3_clip_step_vs_oracle)
    assert num / den < 6e-2 and per[worst] < 1e-1, (what, num / den, worst, per[worst])


@pytest.mark.parametrize("path", ["engine", "module"])
def test_stage2_clip_step_vs_oracle(dev, path):
    """configs[3] in miniature, batch 8: loss and every trainable gradient of the stage-2 step (frozen stage-1 tower forward,
    2E3 fwd+bwd, InfoNCE + relation MSE at step 1000) vs the oracle's autograd; the frozen tower gets no gradient."""
    from hsenet_amd.clip_step import ClipStage2Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    B, step = 8, 1000
    ref, m = _models(dev, 750)
    x, s = R.synthetic_volume((B, 1) + IMG, 751), R.synthetic_slices(B, 752)
    t1, t2 = RC.synthetic_text_features(B, 753), RC.synthetic_text_features(B, 754)
    out_r = ref(x, t1, t2, s, global_step=step)
    out_r["loss"].backward()
    xd, sd_, t1d, t2d = x.to(dev), s.to(dev), t1.to(dev), t2.to(dev)
    if path == "engine":
        eng = ClipStage2Engine(m, B)
        eng.global_step = step
        eng.forward_loss(xd, sd_, t1d, t2d)
        eng.backward()
        torch.cuda.synchronize()
        loss = eng.loss_value()
        assert rel((eng.logits2 + eng.logits2.t()) / 2, out_r["logits"].detach()) < 2e-2
        names = set(eng.params.names)
        assert "logit_scale" in names and "mm_vision_proj.weight" in names and not any(n.startswith("stage1") for n in names)
    else:
        out = m(xd, t1d, t2d, sd_, global_step=step)
        out["loss"].backward()
        loss = float(out["loss"])
        assert rel(out["logits"], out_r["logits"].detach()) < 2e-2
    lr_ = float(out_r["loss"])
    print(f"stage-2 CLIP step ({path}): loss {loss:.5f} vs oracle {lr_:.5f}")
    assert abs(loss - lr_) < 2e-2 * abs(lr_)
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in m.stage1_pretrained_CLIP.parameters())
    _grad_check(list(m.named_parameters()), list(ref.named_parameters()), f"stage-2 CLIP step ({path})")


def _fixture_errors(named, g):
    """(global error over the stored values, worst norm error, worst value error) of gradients against a make_goldens fixture;
    the worst-of lists cover the MAJOR tensors (reference norm > 2 % of the largest)."""
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    mine = {n: p.grad.detach().double().cpu() for n, p in named}
    assert sorted(mine) == sorted(names)
    num = den = 0.0
    big = float(norms.max())
    worst_norm, worst_val = ("", 0.0), ("", 0.0)
    for n, rn in zip(names, norms):
        gm = mine[n]
        major = rn > 2e-2 * big
        if major:
            worst_norm = max(worst_norm, (n, abs(float(gm.norm()) - rn) / rn), key=lambda t: t[1])
        key = f"grad_g/{n}"
        if key not in g.files:
            continue
        ref = torch.from_numpy(g[key]).double()
        rk = f"grad_r/{n}"
        got = gm.reshape(-1, gm.shape[-1])[torch.from_numpy(g[rk])] if rk in g.files else gm.reshape(ref.shape)
        num += float((got - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
        if major and float(ref.norm()) > 0:
            worst_val = max(worst_val, (n, float((got - ref).norm() / ref.norm())), key=lambda t: t[1])
    return (num / den) ** 0.5, worst_norm, worst_val


def test_stage2_clip_step_256_vs_golden(dev, golden):
    """BASELINE configs[3] at FULL geometry (256^3 / 16^3, 12 layers, N = 4097; 2 volumes, global step 1000) against
    tests/golden/clip_c4a_256.npz (oracle/make_goldens.py::clip_c4a -- since round 5 the REFERENCE's own M3DCLIP_stage2.forward +
    autograd run in the build container: loss, logits and the gradient of every trainable vision tensor; the text features its two
    BERT towers produced are stored in the file and are inputs here).
      (1) ClipStage2Engine, bf16 -- the path `bench.py --workload clip_c4a` times: loss pieces within 5e-3, logits within 2e-2.
          Gradients: with random-init towers the two volumes' features nearly coincide (loss ~ ln 2) and the InfoNCE gradients of the
          two samples are almost exactly opposite (cosine of the two token-0 gradients -0.9988, printed), so every parameter gradient
          is a ~20x-cancelled difference: the bf16 policy's ~1 % per-sample error shows up as 15-30 % per tensor.  Checked at that
          level (global 6e-2 over the stored values, major-tensor norms 1e-1, values 0.4) -- a sanity bound, not the parity statement;
      (2) the module path in the exact-f32 parity mode on the same inputs: every stored gradient within 2e-3 (= the f32 mode's ~1e-5
          under the same 20x amplification, with margin) -- THIS pins the step's arithmetic at full size."""
    import hsenet_amd as H
    from hsenet_amd.clip import M3DCLIPVision_stage2
    from hsenet_amd.clip_step import ClipStage2Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    g = golden("clip_c4a_256.npz")
    img, patch, B, step = tuple(int(v) for v in g["img"]), tuple(int(v) for v in g["patch"]), int(g["batch"]), int(g["global_step"])
    m = M3DCLIPVision_stage2(img, patch)
    sd = R.seeded_state_dict(m, int(g["seed_w"]))
    sd["logit_scale"] = torch.tensor(SCALE)
    sd["stage1_pretrained_CLIP.logit_scale"] = torch.tensor(SCALE)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x, s = R.synthetic_volume((B, 1) + img, int(g["seed_x"])).to(dev), R.synthetic_slices(B, int(g["seed_s"])).to(dev)
    t1, t2 = torch.from_numpy(g["text1"]).to(dev), torch.from_numpy(g["text2"]).to(dev)     # the reference's own encode_text outputs
    trainable = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    # ---- (1) the fused engine, bf16
    eng = ClipStage2Engine(m, B)
    eng.global_step = step
    eng.forward_loss(x, s, t1, t2)
    eng.backward()
    torch.cuda.synchronize()
    loss, want = eng.loss_value(), float(g["loss"])
    a = eng.acc.tolist()
    d0 = eng.u2.dx0[:, 0, :].double()
    cos = float(torch.nn.functional.cosine_similarity(d0[0], d0[1], dim=0))
    print(f"configs[3] 256^3 engine: loss {loss:.6f} vs golden {want:.6f} (CL {a[0]:.6f} vs {float(g['loss_cl']):.6f}, relation {a[2]:.6f} vs "
          f"{float(g['loss_relation']):.6f}); per-sample token-0 gradients: cosine {cos:.4f}, |sum| / |one| = "
          f"{float(d0.sum(0).norm() / d0[0].norm()):.3f}")
    assert abs(loss - want) < 5e-3 * abs(want)
    assert abs(a[0] - float(g["loss_cl"])) < 5e-3 * float(g["loss_cl"])
    assert abs(a[2] - float(g["loss_relation"])) < 5e-2 * float(g["loss_relation"]) + 1e-4
    assert rel((eng.logits2 + eng.logits2.t()) / 2, torch.from_numpy(g["logits"])) < 2e-2
    glob, wn, wv = _fixture_errors(trainable, g)
    print(f"configs[3] 256^3 engine gradients (bf16, ~20x cancellation): global {glob:.2e}; major tensors: worst norm {wn[0]} {wn[1]:.2e}, "
          f"worst value {wv[0]} {wv[1]:.2e}")
    assert glob < 6e-2 and wn[1] < 1e-1 and wv[1] < 0.4
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in m.stage1_pretrained_CLIP.parameters())
    del eng
    # ---- (2) the module path, exact f32
    for _, p in trainable:
        p.grad = None
    H.set_precision("f32")
    try:
        out = m(x, t1, t2, s, global_step=step)
        out["loss"].backward()
        torch.cuda.synchronize()
    finally:
        H.set_precision("bf16")
    lf = float(out["loss"])
    glob, wn, wv = _fixture_errors(trainable, g)
    print(f"configs[3] 256^3 module path, f32 parity mode: loss {lf:.7f} vs {want:.7f}; gradients: global {glob:.2e}; worst norm {wn[0]} "
          f"{wn[1]:.2e}, worst value {wv[0]} {wv[1]:.2e}")
    assert abs(lf - want) < 1e-5 * abs(want) + 1e-6
    assert rel(out["logits"], torch.from_numpy(g["logits"])) < 1e-5
    assert glob < 2e-3 and wn[1] < 2e-3 and wv[1] < 2e-3


def test_stage2_clip_engine_trains(dev):
    """a few optimizer steps reduce the loss; the relation weight follows the schedule; nothing syncs on the scale."""
    from hsenet_amd.clip_step import ClipStage2Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    B = 4
    _, m = _models(dev, 770)
    eng = ClipStage2Engine(m, B, lr=3e-4)
    x, s = R.synthetic_volume((B, 1) + IMG, 771).to(dev), R.synthetic_slices(B, 772).to(dev)
    eng.set_text_features(RC.synthetic_text_features(B, 773).to(dev), RC.synthetic_text_features(B, 774).to(dev))
    losses = []
    for i in range(6):
        eng.train_step(x, s)
        losses.append(eng.loss_value())
        assert abs(eng._w - 0.1 * (1 - i / 5000)) < 1e-12
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert eng.global_step == 6 and float(m.logit_scale) != SCALE            # logit_scale is trained (AdamW on the flat buffer)


def test_stage2_clip_engine_refuses_wrong_text_features(dev):
    """the engine hands raw pointers to kernels that read B*d fp32 values: bf16 / fp16 text features (what an autocast BERT tower
    produces) or a wrong shape must raise, not be read out of bounds; set_text_features converts; train_step without either raises."""
    from hsenet_amd import _lib as L
    from hsenet_amd.clip_step import ClipStage2Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    B = 2
    _, m = _models(dev, 990)
    eng = ClipStage2Engine(m, B)
    x, s = R.synthetic_volume((B, 1) + IMG, 991).to(dev), R.synthetic_slices(B, 992).to(dev)
    t1, t2 = RC.synthetic_text_features(B, 993).to(dev), RC.synthetic_text_features(B, 994).to(dev)
    with pytest.raises(L.HsError):
        eng.train_step(x, s)                                       # no text features yet
    with pytest.raises(L.HsError):
        eng.train_step(x, s, t1.bfloat16(), t2)                    # half-size elements: would be an out-of-bounds read
    with pytest.raises(L.HsError):
        eng.train_step(x, s, t1, t2[:, :512].contiguous())         # wrong width
    with pytest.raises(L.HsError):
        eng.train_step(x, s, t1[:1], t2[:1])                       # wrong batch
    with pytest.raises(L.HsError):
        eng.set_text_features(t1, t2[:1])
    eng.train_step(x, s, t1, t2)
    ref_loss = eng.loss_value()
    _, m2 = _models(dev, 990)
    eng2 = ClipStage2Engine(m2, B)
    eng2.set_text_features(t1.to(torch.bfloat16).float().half(), t2.cpu())     # converted (dtype and device) once, here
    assert eng2._txt1.dtype == torch.float32 and eng2._txt2.is_cuda
    eng2.set_text_features(t1, t2)
    eng2.train_step(x, s)
    assert abs(eng2.loss_value() - ref_loss) < 1e-6 * max(1.0, abs(ref_loss))


# ---- the stage-1 CLIP step (CLIP_stage1.py:111-139) and the both-trainable variant (SURVEY section 8(d) "C4-B") ----------------
def _models_s1(dev, seed):
    from hsenet_amd.clip import M3DCLIPVision_stage1
    from oracle import clip_cpu as RC, ref_cpu as R
    ref = RC.M3DCLIPVision_stage1(IMG, PATCH, LAYERS).eval()
    sd = R.seeded_state_dict(ref, seed)
    sd["logit_scale"] = torch.tensor(SCALE)
    ref.load_state_dict(sd)
    m = M3DCLIPVision_stage1(IMG, PATCH, num_layers=LAYERS)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(sd)
    return ref, m.to(dev).eval()


@pytest.mark.parametrize("path", ["engine", "module"])
def test_stage1_clip_step_vs_oracle(dev, path):
    """M3DCLIP_stage1.forward in miniature, batch 8: loss, logits and every gradient (3D-ViT, mm_vision_proj, logit_scale) of
    InfoNCE against frozen text features vs the oracle's autograd, through the fused engine and through the module API."""
    from hsenet_amd.clip_step import ClipStage1Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    B = 8
    ref, m = _models_s1(dev, 1750)
    x, t1 = R.synthetic_volume((B, 1) + IMG, 1751), RC.synthetic_text_features(B, 1753)
    out_r = ref(x, t1)
    out_r["loss"].backward()
    xd, t1d = x.to(dev), t1.to(dev)
    if path == "engine":
        eng = ClipStage1Engine(m, B)
        eng.forward_loss(xd, t1d)
        eng.backward()
        torch.cuda.synchronize()
        loss = eng.loss_value()
        assert rel((eng.logits + eng.logits.t()) / 2, out_r["logits"].detach()) < 2e-2
        assert set(eng.params.names) == set(dict(m.named_parameters()))
    else:
        out = m(xd, t1d)
        out["loss"].backward()
        loss = float(out["loss"])
        assert rel(out["logits"], out_r["logits"].detach()) < 2e-2
    lr_ = float(out_r["loss"])
    print(f"stage-1 CLIP step ({path}): loss {loss:.5f} vs oracle {lr_:.5f}")
    assert abs(loss - lr_) < 2e-2 * abs(lr_)
    _grad_check(list(m.named_parameters()), list(ref.named_parameters()), f"stage-1 CLIP step ({path})")


def test_stage1_clip_engine_trains_and_refuses_frozen_models(dev):
    from hsenet_amd import _lib as L
    from hsenet_amd.clip_step import ClipStage1Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    B = 4
    _, m = _models_s1(dev, 1770)
    eng = ClipStage1Engine(m, B, lr=3e-4)
    x = R.synthetic_volume((B, 1) + IMG, 1771).to(dev)
    with pytest.raises(L.HsError):
        eng.train_step(x)                                          # no text features yet
    eng.set_text_features(RC.synthetic_text_features(B, 1773).to(dev).half())       # an autocast text tower hands over fp16: converted
    losses = []
    for _ in range(6):
        eng.train_step(x)
        losses.append(eng.loss_value())
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert eng.global_step == 6 and float(m.logit_scale) != SCALE
    _, frozen = _models_s1(dev, 1771)
    frozen.requires_grad_(False)
    with pytest.raises(L.HsError):
        ClipStage1Engine(frozen, B)


def test_both_trainable_clip_step_vs_oracle(dev):
    """SURVEY section 8(d) C4-B in miniature, batch 8: ONE iteration trains the stage-1 CLIP (its own InfoNCE, its own logit_scale)
    and the stage-2 CLIP (InfoNCE + relation MSE against the DETACHED stage-1 logits) on a shared stage-1 forward: both losses and
    every gradient of both models vs the oracle's autograd of loss_stage1 + loss_stage2; then both optimizers step."""
    from hsenet_amd.clip_step import ClipStage2Engine
    from oracle import clip_cpu as RC, ref_cpu as R
    B, step = 8, 1000
    ref, m = _models(dev, 1850)
    ref.stage1_pretrained_CLIP.requires_grad_(True)
    m.stage1_pretrained_CLIP.requires_grad_(True)
    with torch.no_grad():                                           # a stage-1 logit_scale of its own: the two models' scales differ
        ref.stage1_pretrained_CLIP.logit_scale.fill_(2.0)
        m.stage1_pretrained_CLIP.logit_scale.fill_(2.0)
    x, s = R.synthetic_volume((B, 1) + IMG, 1851), R.synthetic_slices(B, 1852)
    t1, t2 = RC.synthetic_text_features(B, 1853), RC.synthetic_text_features(B, 1854)
    total, l1, l2 = RC.c4b_losses(ref, x, t1, t2, s, step)
    total.backward()
    eng = ClipStage2Engine(m, B, train_stage1=True)
    eng.global_step = step
    eng.forward_loss(x.to(dev), s.to(dev), t1.to(dev), t2.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    mine1, mine2 = eng.s1.loss_value(), eng.loss_value()
    print(f"both-trainable CLIP step: stage-1 loss {mine1:.5f} vs {float(l1):.5f}, stage-2 loss {mine2:.5f} vs {float(l2):.5f}")
    assert abs(mine1 - float(l1)) < 2e-2 * abs(float(l1)) and abs(mine2 - float(l2)) < 2e-2 * abs(float(l2))
    assert not set(eng.params.names) & {"stage1_pretrained_CLIP." + n for n in eng.s1.params.names}
    s1_ref = [(n, p) for n, p in ref.named_parameters() if n.startswith("stage1_pretrained_CLIP.")]
    s2_ref = [(n, p) for n, p in ref.named_parameters() if not n.startswith("stage1_pretrained_CLIP.")]
    _grad_check([(n, p) for n, p in m.named_parameters() if n.startswith("stage1_pretrained_CLIP.")], s1_ref, "C4-B stage-1 half")
    _grad_check([(n, p) for n, p in m.named_parameters() if not n.startswith("stage1_pretrained_CLIP.")], s2_ref, "C4-B stage-2 half")
    before = [float(m.logit_scale), float(m.stage1_pretrained_CLIP.logit_scale)]
    eng.train_step(x.to(dev), s.to(dev), t1.to(dev), t2.to(dev))
    torch.cuda.synchronize()
    assert eng.global_step == step + 1 and eng.s1.global_step == 1
    assert float(m.logit_scale) != before[0] and float(m.stage1_pretrained_CLIP.logit_scale) != before[1]
    assert bool(torch.isfinite(eng.flat).all()) and bool(torch.isfinite(eng.s1.flat).all())
