"""The train-step engines (hsenet_amd/engine.py — the path bench.py times) against the CPU oracle: DualEngine =
BASELINE configs[2] (dual encoders + two Spatial Packers), FlatEncoder = configs[1]; forward output, every parameter
gradient, and the optimizer step (HF-Trainer semantics: no decay on biases / LayerNorm, global-norm clip) vs torch.optim.AdamW."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_BF16 = 1e-2      # forward, bf16 MFMA inputs (measured 4-7e-3; the oracle's own bf16-autocast emulation differs from fp32 by 4.5-5.3e-3)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


class Cfg:
    vision_tower = "vit_stage2_dual_encoders"
    vision_select_layer, vision_select_feature, image_channel = -1, "patch", 1
    remain_2d3d_ViT_type = "dual_vits"

    def __init__(self, img, patch):
        self.image_size, self.patch_size = img, patch


IMG, PATCH, GRID, KERN, OUT = (8, 64, 64), (4, 16, 16), (2, 4, 4), (1, 2, 2), 256
PARGS = (IMG, PATCH, 768, OUT, "mlp", 2)


def _shrink(tower, layers):
    for t in (tower.vision_tower_stage1, tower.vision_tower_stage2):
        del t.blocks[layers:]
    return tower


def _oracle(B, seed, layers=2):
    from oracle import ref_cpu as R
    tower = _shrink(R.ViT3DTower_dual_encoders(Cfg(IMG, PATCH)), layers).eval()
    p1 = R.VisualPacker_3d_phi_v3(*PARGS, grid=GRID, kernel_size=KERN).eval()
    p2 = R.VisualPacker_3d_phi_v3(*PARGS, grid=GRID, kernel_size=KERN).eval()
    sds = [R.seeded_state_dict(m, seed + i) for i, m in enumerate((tower, p1, p2))]
    for m, sd in zip((tower, p1, p2), sds):
        m.load_state_dict(sd)
    x, s = R.synthetic_volume((B, 1) + IMG, seed + 10), R.synthetic_slices(B, seed + 11)
    return tower, p1, p2, sds, x, s


def _mine(sds, dev, layers=2, train=False):
    import hsenet_amd as H
    tower = _shrink(H.build_vision_tower(Cfg(IMG, PATCH)), layers)
    p1 = H.VisualPacker_3d_phi_v3(*PARGS, grid=GRID, kernel_size=KERN)
    p2 = H.VisualPacker_3d_phi_v3(*PARGS, grid=GRID, kernel_size=KERN)
    for m, sd in zip((tower, p1, p2), sds):
        m.load_state_dict(sd)
        m.to(dev)
        m.train(train)
    return tower, p1, p2


def _named(tower, p1, p2):
    return ([("vision_tower." + n, p) for n, p in tower.named_parameters()] + [("mm_projector." + n, p) for n, p in p1.named_parameters()]
            + [("mm_projector2." + n, p) for n, p in p2.named_parameters()])


def test_dual_engine_vs_oracle(dev):
    """configs[2] in miniature (grid (2,4,4), kernel (1,2,2), L=2, batch 3): engine output and ALL parameter gradients of
    L = mean(out^2) vs the oracle's autograd; then one optimizer step vs torch.optim.AdamW with HF Trainer's grouping + clip."""
    from hsenet_amd.engine import DualEngine, decays
    from oracle import ref_cpu as R
    from test_gpu_parity import grad_errs
    B = 3
    tr, p1r, p2r, sds, x, s = _oracle(B, 900)
    out_r = R.encode_images(tr, p1r, p2r, x, s)
    (out_r * out_r).mean().backward()
    tower, p1, p2 = _mine(sds, dev)
    eng = DualEngine(tower, p1, p2, B, lr=1e-3, weight_decay=0.1, max_grad_norm=1.0)
    assert eng.params.total >= sum(p.numel() for _, p in _named(tower, p1, p2))
    out = eng.forward(x.to(dev), s.to(dev))
    assert tuple(out.shape) == (B, 2 * p1.proj_out_num, OUT) == tuple(out_r.shape)
    e = rel(out, out_r.detach())
    print(f"DualEngine fwd: {e:.2e}")
    assert e < TOL_BF16
    from hsenet_amd import _lib as L, functional as Fn
    eng.loss.zero_()
    L.call("hs_mse_grad", out.data_ptr(), eng.dout.data_ptr(), eng.loss.data_ptr(), out.numel(), Fn._s())
    eng.backward(eng.dout)
    torch.cuda.synchronize()
    lr_ = float((out_r * out_r).mean().detach())
    assert abs(float(eng.loss) - lr_) < 3e-2 * lr_
    errs = grad_errs(_named(tower, p1, p2), _named(tr, p1r, p2r))
    worst = max(errs, key=errs.get)
    print(f"DualEngine backward worst grad err: {worst} {errs[worst]:.2e} over {len(errs)} tensors")
    assert errs[worst] < 5e-2, (worst, errs[worst])
    for bad in (eng.dout.bfloat16(), eng.dout[:, :-1].contiguous(), eng.dout.transpose(1, 2)):
        with pytest.raises(L.HsError):            # dout is read as B*256*hidden fp32 values through its raw pointer
            eng.backward(bad)

    # ---- optimizer: same gradients into torch.optim.AdamW the way HF Trainer builds it
    ref_named = _named(tr, p1r, p2r)
    mine_named = dict(_named(tower, p1, p2))
    dec = {}
    for pre, mod in (("vision_tower.", tr), ("mm_projector.", p1r), ("mm_projector2.", p2r)):
        dec.update({pre + n: v for n, v in decays(mod).items()})
    assert not dec["vision_tower.vision_tower_stage1.norm.weight"] and not dec["mm_projector.proj_mpls.0.bias"]
    assert dec["vision_tower.vision_tower_stage1.cls_token"] and dec["vision_tower.vision_tower_stage2.blocks.0.attn.qkv.weight"]
    for n, p in ref_named:
        p.grad = mine_named[n].grad.detach().cpu().clone()
    opt = torch.optim.AdamW([{"params": [p for n, p in ref_named if dec[n]], "weight_decay": 0.1},
                             {"params": [p for n, p in ref_named if not dec[n]], "weight_decay": 0.0}], lr=1e-3)
    before = {n: p.detach().clone() for n, p in ref_named}
    gn = torch.nn.utils.clip_grad_norm_([p for _, p in ref_named], 1.0)
    opt.step()
    eng.optimizer_step()
    torch.cuda.synchronize()
    gn_mine = float(eng.params.gsq.sqrt())
    assert abs(gn_mine - float(gn)) < 1e-3 * float(gn), (gn_mine, float(gn))
    for n, p in ref_named:
        moved = float((p.detach() - before[n]).norm())
        diff = float((mine_named[n].detach().cpu() - p.detach()).norm())
        assert diff <= 1e-3 * moved + 1e-7, (n, diff, moved)
    print(f"optimizer step matches torch AdamW (+clip at |g| = {float(gn):.3f}) on {len(ref_named)} tensors")


def test_dual_engine_f16_train_step_vs_oracle(dev):
    """round 5: DualEngine(precision='f16') -- every 16-bit tensor IEEE fp16, default power-of-two loss scale -- against the fp32 oracle:
    forward inside 1e-3, every gradient (after unscaling) ~4x closer than the bf16 engine's, the AdamW step equal to torch.optim.AdamW
    on the UNSCALED gradients (the scale is divided out inside hs_adamw_masked, the clip acts on the true norm), and an overflowing
    loss scale leaves parameters and moments untouched (non-finite norm -> skipped step)."""
    from hsenet_amd.engine import DualEngine, decays, default_loss_scale
    from oracle import ref_cpu as R
    from test_gpu_parity import grad_errs
    B = 3
    tr, p1r, p2r, sds, x, s = _oracle(B, 900)
    out_r = R.encode_images(tr, p1r, p2r, x, s)
    (out_r * out_r).mean().backward()
    worst = {}
    for prec in ("bf16", "f16"):
        tower, p1, p2 = _mine(sds, dev)
        eng = DualEngine(tower, p1, p2, B, lr=1e-3, weight_decay=0.1, max_grad_norm=1.0, precision=prec)
        if prec == "f16":
            assert eng.loss_scale == default_loss_scale(B * 2 * p1.proj_out_num * OUT) == 2.0 ** 13 and eng.patches.dtype == torch.float16
            assert default_loss_scale(53 * 256 * 3072) == 2.0 ** 25 and default_loss_scale(1) == 1.0
        eng.train_step(x.to(dev), s.to(dev), optimizer=False)
        torch.cuda.synchronize()
        e = rel(eng.out, out_r.detach())
        assert eng.unscaled_grad().shape == eng.grad.shape
        if prec == "f16":
            eng.grad.div_(eng.loss_scale)        # in place, so that the parameters' .grad views hold the true gradient for the comparison
        errs = grad_errs(_named(tower, p1, p2), _named(tr, p1r, p2r))
        w = max(errs, key=errs.get)
        worst[prec] = (e, errs[w], w)
        print(f"DualEngine [{prec}] vs the fp32 oracle: forward {e:.2e}; worst gradient {w} {errs[w]:.2e}; loss {float(eng.loss):.6f} / "
              f"{float((out_r * out_r).mean()):.6f}")
        if prec == "f16":
            eng.grad.mul_(eng.loss_scale)        # back to what the optimizer step expects
            assert abs(float(eng.loss) - float((out_r * out_r).mean())) < 2e-3 * float((out_r * out_r).mean())
    assert worst["f16"][0] < 1e-3 and worst["f16"][1] < 1e-2 and worst["f16"][1] < 0.35 * worst["bf16"][1]
    # ---- optimizer step on the scaled gradient == torch AdamW (+ clip) on the unscaled one
    ref_named = _named(tr, p1r, p2r)
    mine_named = dict(_named(tower, p1, p2))
    dec = {}
    for pre, mod in (("vision_tower.", tr), ("mm_projector.", p1r), ("mm_projector2.", p2r)):
        dec.update({pre + n: v for n, v in decays(mod).items()})
    for n, p in ref_named:
        p.grad = (mine_named[n].grad.detach() / eng.loss_scale).cpu().clone()
    opt = torch.optim.AdamW([{"params": [p for n, p in ref_named if dec[n]], "weight_decay": 0.1},
                             {"params": [p for n, p in ref_named if not dec[n]], "weight_decay": 0.0}], lr=1e-3)
    before = {n: p.detach().clone() for n, p in ref_named}
    gn = torch.nn.utils.clip_grad_norm_([p for _, p in ref_named], 1.0)
    opt.step()
    eng.optimizer_step()
    torch.cuda.synchronize()
    for n, p in ref_named:
        moved = float((p.detach() - before[n]).norm())
        diff = float((mine_named[n].detach().cpu() - p.detach()).norm())
        assert diff <= 1e-3 * moved + 1e-7, (n, diff, moved)
    print(f"fp16 engine: optimizer step on the loss-scaled gradient matches torch AdamW on the unscaled one (clip at |g| = {float(gn):.3f})")
    # ---- an absurd loss scale overflows fp16: the gradient norm is not finite and the step is skipped
    eng.loss_scale = 2.0 ** 40
    snap, m_snap = eng.flat.clone(), eng.params.m.clone()
    eng.train_step(x.to(dev), s.to(dev))
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(eng.params.gsq).all()) and torch.equal(eng.flat, snap) and torch.equal(eng.params.m, m_snap)
    with pytest.raises(Exception):
        DualEngine(*_mine(sds, dev), B, precision="f32")
    with pytest.raises(Exception):
        DualEngine(*_mine(sds, dev), B, precision="bf16", loss_scale=8.0)


def test_f16_engine_tracks_the_fp32_trajectory(dev):
    """ten optimizer steps of configs[2] in miniature (eval-mode dropout sites, lr 1e-3, clip 1.0, HF decay groups) from the same weights:
    the fp32 CPU oracle + torch.optim.AdamW is the trajectory; the fp16 engine (loss scale in, loss scale out, every step) stays several
    times closer to it than the bf16 engine does -- in parameters and in the loss at every step -- and no step is skipped."""
    from hsenet_amd.engine import DualEngine, decays
    from oracle import ref_cpu as R
    B, steps = 2, 10
    tr, p1r, p2r, sds, x, s = _oracle(B, 960)
    ref_named = _named(tr, p1r, p2r)
    dec = {}
    for pre, mod in (("vision_tower.", tr), ("mm_projector.", p1r), ("mm_projector2.", p2r)):
        dec.update({pre + n: v for n, v in decays(mod).items()})
    opt = torch.optim.AdamW([{"params": [p for n, p in ref_named if dec[n]], "weight_decay": 0.1},
                             {"params": [p for n, p in ref_named if not dec[n]], "weight_decay": 0.0}], lr=1e-3)
    start = {n: p.detach().clone() for n, p in ref_named}
    ref_losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        out = R.encode_images(tr, p1r, p2r, x, s)
        loss = (out * out).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for _, p in ref_named], 1.0)
        opt.step()
        ref_losses.append(float(loss))
    moved = sum(float((p.detach() - start[n]).pow(2).sum()) for n, p in ref_named) ** 0.5
    dist, lerrs = {}, {}
    for prec in ("bf16", "f16"):
        tower, p1, p2 = _mine(sds, dev)
        eng = DualEngine(tower, p1, p2, B, lr=1e-3, weight_decay=0.1, max_grad_norm=1.0, precision=prec)
        losses = []
        for _ in range(steps):
            eng.train_step(x.to(dev), s.to(dev))
            losses.append(float(eng.loss))
        torch.cuda.synchronize()
        assert eng.step_no == steps and bool(torch.isfinite(eng.flat).all()) and bool(torch.isfinite(eng.params.gsq).all())
        mine = dict(_named(tower, p1, p2))
        d = sum(float((mine[n].detach().cpu() - p.detach()).pow(2).sum()) for n, p in ref_named) ** 0.5
        dist[prec] = d / moved
        lerr = max(abs(a - b) / b for a, b in zip(losses, ref_losses))
        print(f"DualEngine [{prec}] after {steps} AdamW steps: |theta - theta_fp32| / |theta_fp32 - theta_0| = {dist[prec]:.2e}; worst loss error {lerr:.1e}; "
              f"loss {losses[0]:.5f} -> {losses[-1]:.5f} (oracle {ref_losses[0]:.5f} -> {ref_losses[-1]:.5f})")
        lerrs[prec] = lerr
        assert losses[-1] < losses[0] and lerr < 6e-2
    assert lerrs["f16"] <= lerrs["bf16"]                  # (the loss falls 40x in these ten steps: late-step RELATIVE loss errors are amplified for both)
    # (AdamW's sign-like early steps amplify tiny gradient differences on near-zero-gradient coordinates, so the distance is not 1e-3)
    assert dist["f16"] < 0.5 * dist["bf16"] and dist["f16"] < 0.1


def test_dual_engine_matches_module_path(dev):
    """engine (fused drivers, flat buffers) == module API (per-op autograd Functions) on the same weights: same kernels
    underneath, so only summation order / epilogue fusion differ."""
    import hsenet_amd as H
    from hsenet_amd.engine import DualEngine
    from hsenet_amd import _lib as L, functional as Fn
    B = 2
    _, _, _, sds, x, s = _oracle(B, 920)
    tower_m, p1m, p2m = _mine(sds, dev)
    out_m = H.encode_images(tower_m, p1m, p2m, x.to(dev), s.to(dev))
    (out_m * out_m).mean().backward()
    tower, p1, p2 = _mine(sds, dev)
    eng = DualEngine(tower, p1, p2, B)
    out = eng.forward(x.to(dev), s.to(dev))
    L.call("hs_mse_grad", out.data_ptr(), eng.dout.data_ptr(), None, out.numel(), Fn._s())
    eng.backward(eng.dout)
    torch.cuda.synchronize()
    assert rel(out, out_m.detach()) < 8e-3      # the engine's packer fuses GELU into the GEMM epilogue (bf16-rounded pre-activation)
    ref = dict(_named(tower_m, p1m, p2m))
    worst = 0.0
    for n, p in _named(tower, p1, p2):
        g = ref[n].grad
        e = float((p.grad - g).norm() / g.norm().clamp_min(1e-12))
        worst = max(worst, e)
        assert e < 2e-2 or float(g.norm()) < 1e-6, (n, e)
    print(f"engine vs module path: worst grad diff {worst:.2e}")


def test_dual_engine_train_mode_and_state_dict(dev):
    """train(): the hard-coded Dropout(0.1) sites fire (vit.py:46-47, spatial_pooling_projector.py:58-59) and replay under
    torch.manual_seed; adopting the parameters keeps state_dict()/load_state_dict() working; steps reduce the loss."""
    from hsenet_amd.engine import DualEngine
    B = 2
    _, _, _, sds, x, s = _oracle(B, 940)
    tower, p1, p2 = _mine(sds, dev, train=True)
    keys = list(tower.state_dict().keys())
    eng = DualEngine(tower, p1, p2, B, lr=1e-3)
    assert list(tower.state_dict().keys()) == keys
    xd, sd_ = x.to(dev), s.to(dev)
    torch.manual_seed(5)
    a = eng.forward(xd, sd_).clone()
    torch.manual_seed(5)
    b = eng.forward(xd, sd_).clone()
    torch.manual_seed(6)
    c = eng.forward(xd, sd_).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    for m in (tower, p1, p2):
        m.eval()
    d = eng.forward(xd, sd_).clone()
    assert rel(a, d) > 1e-3                      # dropout really changed the function
    losses = []
    for _ in range(4):
        eng.train_step(xd, sd_)
        losses.append(float(eng.loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # load_state_dict writes through to the flat buffer
    tower.load_state_dict(sds[0])
    off = eng.params.offset("vision_tower_stage1.norm.weight")
    assert torch.equal(eng.flat[off:off + 768].cpu(), sds[0]["vision_tower_stage1.norm.weight"])


def test_flat_encoder_stage2_vs_oracle(dev):
    """FlatEncoder also drives the 2E3 encoder alone (the trainable half of the stage-2 CLIP step)."""
    import hsenet_amd as H
    from hsenet_amd.engine import FlatEncoder
    from hsenet_amd import _lib as L, functional as Fn
    from oracle import ref_cpu as R
    from test_gpu_parity import grad_errs
    kw = dict(in_channels=1, img_size=IMG, patch_size=PATCH, num_layers=2, pos_embed="perceptron", classification=True)
    B = 2
    ref = R.ViT_stage2(**kw).eval()
    sd = R.seeded_state_dict(ref, 960)
    ref.load_state_dict(sd)
    x, s = R.synthetic_volume((B, 1) + IMG, 961), R.synthetic_slices(B, 962)
    yr, _ = ref(x, s)
    (yr * yr).mean().backward()
    m = H.ViT_stage2(**kw).eval()
    m.load_state_dict(sd)
    eng = FlatEncoder(m.to(dev), B)
    y = eng.forward(x.to(dev), s.to(dev))
    assert rel(y, yr.detach()) < TOL_BF16
    L.call("hs_mse_grad", y.data_ptr(), eng.dy.data_ptr(), None, y.numel(), Fn._s())
    eng.backward(eng.dy)
    torch.cuda.synchronize()
    errs = grad_errs(m.named_parameters(), ref.named_parameters())
    worst = max(errs, key=errs.get)
    print(f"FlatEncoder(ViT_stage2) worst grad err: {worst} {errs[worst]:.2e}")
    assert errs[worst] < 5e-2, (worst, errs[worst])
    with pytest.raises(L.HsError):
        eng.forward(x.to(dev).transpose(3, 4), s.to(dev))          # non-contiguous volume: refused, not silently mis-read
    # backward entry point: the raw pointer is read as B*N*d fp32 values
    for bad in (eng.dy.bfloat16(), eng.dy[:, :-1].contiguous(), eng.dy.double(), eng.dy.cpu()):
        with pytest.raises(L.HsError):
            eng.backward(bad)
    with pytest.raises(L.HsError):
        eng.forward(x.to(dev), s.to(dev)[:, :16].contiguous())      # slice features of the wrong shape
    with pytest.raises(L.HsError):
        eng.forward(x.to(dev), s.to(dev).bfloat16())


def test_mrg_step_engine_vs_oracle(dev):
    """BASELINE configs[4] in miniature: frozen dual towers + trainable packers -> splice into the token embeddings of a tiny HF
    Phi-3 decoder (the out-of-scope LLM, the same weights on both sides) -> LM loss; packer gradients vs the oracle's autograd,
    no gradient for the towers, and one AdamW step moves only the packers."""
    from transformers import Phi3Config, Phi3ForCausalLM
    from hsenet_amd.mrg_step import MRGStepEngine
    from oracle import ref_cpu as R, clip_cpu as RC
    from test_gpu_parity import grad_errs
    B, S = 2, 24
    tr, p1r, p2r, sds, x, s = _oracle(B, 980)
    tr.requires_grad_(False)
    cfg = Phi3Config(hidden_size=OUT, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=500,
                     pad_token_id=0, bos_token_id=1, eos_token_id=2, attn_implementation="eager")
    torch.manual_seed(3)
    llm_r = Phi3ForCausalLM(cfg).eval()
    llm_r.requires_grad_(False)
    ids = torch.randint(3, 500, (B, S))
    labels = ids.clone()
    labels[:, :1 + 16] = -100
    emb = llm_r.get_input_embeddings()(ids).detach()
    feats_r = R.encode_images(tr, p1r, p2r, x, s)                              # (B, 16, OUT): 2 packers x 8 cells
    loss_r = llm_r(inputs_embeds=RC.splice_image_features(emb, feats_r), labels=labels).loss
    loss_r.backward()
    tower, p1, p2 = _mine(sds, dev)
    eng = MRGStepEngine(tower, p1, p2, B, lr=1e-3)
    llm = Phi3ForCausalLM(cfg).eval()
    llm.load_state_dict(llm_r.state_dict())
    llm = llm.to(dev)
    llm.requires_grad_(False)
    before = {n: p.detach().clone() for n, p in _named(tower, p1, p2)}
    eng.train_step(x.to(dev), s.to(dev), emb.to(dev), lambda e: llm(inputs_embeds=e, labels=labels.to(dev)).loss)
    torch.cuda.synchronize()
    assert abs(float(eng.last_loss) - float(loss_r)) < 2e-3 * abs(float(loss_r)), (float(eng.last_loss), float(loss_r))
    mine = [(n, p) for n, p in _named(tower, p1, p2) if n.startswith("mm_projector")]
    ref = [(n, p) for n, p in _named(tr, p1r, p2r) if n.startswith("mm_projector")]
    errs = grad_errs(mine, ref)
    worst = max(errs, key=errs.get)
    print(f"MRG step: loss {float(eng.last_loss):.4f} vs {float(loss_r):.4f}; worst packer grad err {worst} {errs[worst]:.2e}")
    assert errs[worst] < 5e-2, (worst, errs[worst])
    for n, p in _named(tower, p1, p2):
        moved = float((p.detach() - before[n]).abs().max())
        assert (moved > 0) == n.startswith("mm_projector"), (n, moved)           # towers frozen, packers trained


def test_forward_backward_capture_into_a_hip_graph(dev):
    """the C ABI never allocates, frees or synchronises, so a sequence of calls can be captured in a hipGraph.  The
    fused forward + backward of an encoder (hs_embed_* + hs_vit_blocks_* + the loss gradient: ~150 launches) is captured once and
    replayed on NEW input written into the captured buffer; outputs and gradients must equal the eager run on that input."""
    import hsenet_amd as H
    from hsenet_amd import _lib as L, functional as Fn
    from hsenet_amd.engine import FlatEncoder
    from oracle import ref_cpu as R
    kw = dict(in_channels=1, img_size=IMG, patch_size=PATCH, num_layers=2, pos_embed="perceptron", classification=True)
    ref = R.ViT_stage1(**kw)
    m = H.ViT_stage1(**kw).eval()
    m.load_state_dict(R.seeded_state_dict(ref, 970))
    eng = FlatEncoder(m.to(dev), 2)
    xa, xb = R.synthetic_volume((2, 1) + IMG, 971).to(dev), R.synthetic_volume((2, 1) + IMG, 972).to(dev)

    def fwd_bwd(x):
        y = eng.forward(x)
        L.call("hs_mse_grad", y.data_ptr(), eng.dy.data_ptr(), None, y.numel(), Fn._s())
        eng.backward(eng.dy)
        return y

    y_b = fwd_bwd(xb).clone()                                      # eager reference on input B (also warms every lazy allocation up)
    g_b = eng.grad.clone()
    x = xa.clone()
    fwd_bwd(x)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = fwd_bwd(x)
    x.copy_(xb)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, y_b)
    # weight gradients are float-atomic sums (split-K): equal up to the order of the adds
    assert rel(eng.grad, g_b) < 1e-5


class _FixedMask(torch.nn.Module):
    """stands in for an nn.Dropout of the oracle: multiplies by a given keep-mask (already scaled by 1 / (1 - p))."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask.view(x.shape)


def _philox_mask(n, seed, offset, dev, p=0.1):
    """the engine's Philox keep-mask of a dropout site with n elements: hs_dropout applied to ones (values 0 or 1 / (1 - p))."""
    from hsenet_amd import _lib as L, functional as Fn
    ones = torch.ones(n, dtype=torch.float32, device=dev)
    out = torch.empty_like(ones)
    L.call("hs_dropout", ones.data_ptr(), out.data_ptr(), n, float(p), int(seed), int(offset), Fn._s())
    return out.cpu()


def test_dual_engine_train_mode_step_vs_oracle_with_the_same_masks(dev):
    """bench.py times the engines in train(): the four hard-coded Dropout(0.1) sites of the scorer (vit.py:46-47) and the four of the two
    packers (spatial_pooling_projector.py:58-59) fire.  The engine's masks are a pure function of (seed, offset, element index), so they
    can be exported (hs_dropout on ones) and handed to the oracle in place of its nn.Dropout modules: the train-mode forward output and
    EVERY parameter gradient are then comparable value by value, not just statistically."""
    from hsenet_amd.engine import DualEngine
    from hsenet_amd import _lib as L, functional as Fn
    from oracle import ref_cpu as R
    from test_gpu_parity import grad_errs
    B = 3
    tr, p1r, p2r, sds, x, s = _oracle(B, 980)
    tower, p1, p2 = _mine(sds, dev, train=True)
    eng = DualEngine(tower, p1, p2, B)
    torch.manual_seed(11)
    out = eng.forward(x.to(dev), s.to(dev)).clone()
    L.call("hs_mse_grad", out.data_ptr(), eng.dout.data_ptr(), None, out.numel(), Fn._s())
    eng.backward(eng.dout)
    torch.cuda.synchronize()
    # export the masks of this forward call
    e, Np, d = eng.u2.edesc, eng.u2.Np, eng.u2.d
    assert abs(float(e.drop_p) - 0.1) < 1e-7
    sga = tr.vision_tower_stage2.slice_guided_attention
    sga.dropout = _FixedMask(_philox_mask(B * Np * 32, e.seed[0], e.offset[0], dev))               # probabilities [B][Np][32]
    sga.dropout_2 = _FixedMask(_philox_mask(B * Np * d, e.seed[1], e.offset[1], dev))              # attention output [B][Np][d]
    for unit, pr in ((eng.p1, p1r), (eng.p2, p2r)):
        pd, cells = unit.desc, unit.cells
        nwin = pd.kd * pd.kh * pd.kw
        m = _philox_mask(B * cells * 64, pd.seed[0], pd.offset[0], dev).view(B * cells, 64)[:, :nwin]   # lane j of cell c: index c * 64 + j
        pr.resolution_attention.dropout = _FixedMask(m.contiguous())
        pr.resolution_attention.dropout_2 = _FixedMask(_philox_mask(B * cells * pd.d, pd.seed[1], pd.offset[1], dev))
    for mref in (tr, p1r, p2r):
        mref.train()
    out_r = R.encode_images(tr, p1r, p2r, x, s)
    (out_r * out_r).mean().backward()
    e_fwd = rel(out, out_r.detach())
    print(f"train-mode DualEngine forward vs the oracle with the same masks: {e_fwd:.2e}")
    assert e_fwd < TOL_BF16
    keep = float((sga.dropout.mask > 0).float().mean())
    assert 0.88 < keep < 0.92                                          # the exported mask is a 10 % dropout mask
    errs = grad_errs(_named(tower, p1, p2), _named(tr, p1r, p2r))
    worst = max(errs, key=errs.get)
    print(f"train-mode backward worst grad err: {worst} {errs[worst]:.2e} over {len(errs)} tensors")
    assert errs[worst] < 5e-2, (worst, errs[worst])
    # and the masks matter: the oracle with OTHER masks (the next Philox offsets) is measurably further from the engine's output
    sga.dropout = _FixedMask(_philox_mask(B * Np * 32, e.seed[0], e.offset[0] + 7919, dev))
    sga.dropout_2 = _FixedMask(_philox_mask(B * Np * d, e.seed[1], e.offset[1] + 7919, dev))
    for unit, pr in ((eng.p1, p1r), (eng.p2, p2r)):
        pd, cells = unit.desc, unit.cells
        nwin = pd.kd * pd.kh * pd.kw
        m = _philox_mask(B * cells * 64, pd.seed[0], pd.offset[0] + 7919, dev).view(B * cells, 64)[:, :nwin]
        pr.resolution_attention.dropout = _FixedMask(m.contiguous())
        pr.resolution_attention.dropout_2 = _FixedMask(_philox_mask(B * cells * pd.d, pd.seed[1], pd.offset[1] + 7919, dev))
    e_other = rel(out, R.encode_images(tr, p1r, p2r, x, s).detach())
    print(f"... with other masks: {e_other:.2e}")
    assert e_other > 1.25 * e_fwd


def test_default_bench_batch_is_consistent(dev):
    """the configuration bench.py times by default (configs[2], 53 volumes per GPU, ~197 GB of HBM; the largest batch that fits when the
    GPU is not empty): the forward output of the first and of the last volume is bit-identical to the same volume run alone (row tiles
    and (batch, head) blocks do not depend on the batch), and for a batch of identical volumes the mean-square loss and its gradient
    equal the 1-volume ones (linearity)."""
    import bench as Bn
    from hsenet_amd import _lib as L, functional as Fn
    free = torch.cuda.mem_get_info(dev)[0]
    B = next((b for b in (Bn.DEFAULT_BATCH, 31, 20, 8, 4) if free > (3.85 * b + 12) * 1e9), None)
    if B is None:
        pytest.skip("less than 28 GB of free HBM")
    geom = Bn.GEOM["dual256"]
    rng = np.random.default_rng(7)
    vol1 = torch.from_numpy(rng.random((1, 1) + geom["img"], dtype=np.float32)).to(dev)
    s1 = torch.from_numpy(rng.standard_normal((1, 32, 768), dtype=np.float32)).to(dev)

    def run(b):
        eng, _, _ = Bn.build_engine("dual256", geom, b, dev)
        for m in (eng.tower, eng.p1.packer, eng.p2.packer):
            m.eval()
        out = eng.forward(vol1.expand(b, -1, -1, -1, -1).contiguous(), s1.expand(b, -1, -1).contiguous())
        eng.loss.zero_()
        L.call("hs_mse_grad", out.data_ptr(), eng.dout.data_ptr(), eng.loss.data_ptr(), out.numel(), Fn._s())
        eng.backward(eng.dout)
        torch.cuda.synchronize()
        res = out[0].clone(), out[b - 1].clone(), eng.grad.clone(), float(eng.loss)
        del eng
        torch.cuda.empty_cache()
        return res
    o1, _, g1, l1 = run(1)
    o0, ol, gB, lB = run(B)
    print(f"batch {B}: loss {lB:.8f} vs {l1:.8f}; gradient difference {rel(gB, g1):.2e}")
    assert torch.equal(o0, o1) and torch.equal(ol, o1)
    assert abs(lB - l1) < 1e-5 * abs(l1)
    assert rel(gB, g1) < 2e-3                                          # measured 4.6e-4 at 53 volumes (fp32 summation order over the batch)


def test_mlp_recompute_is_bit_identical_and_smaller(dev):
    """DualEngine(recompute_mlp=True) / hs_vit_desc.save_for_bwd = 2: gelu(z) and gelu'(z) are not kept per layer, the backward re-runs fc1
    with its GELU epilogue from the saved LN2 output (same kernel, same operands) — output, gradients and the parameters after two
    optimizer steps equal the default engine's (forward bit for bit, gradients to the float-atomic noise of two runs); the encoder workspace shrinks by the two [rows][mlp] arrays of L - 1 layers."""
    import ctypes as C
    from hsenet_amd import _lib as L, functional as Fn
    from hsenet_amd.engine import DualEngine
    B, layers = 3, 3
    _, _, _, sds, x, s = _oracle(B, 4100, layers)
    xd, sd_ = x.to(dev), s.to(dev)
    engs = []
    for rec in (False, True):
        tower, p1, p2 = _mine(sds, dev, layers)
        engs.append(DualEngine(tower, p1, p2, B, lr=1e-3, recompute_mlp=rec))
    a, b = engs
    assert b.u1.desc.save_for_bwd == 2 and a.u1.desc.save_for_bwd == 1
    N = a.u1.N
    assert a.u1.ws_bytes - b.u1.ws_bytes >= 2 * (layers - 1) * B * N * 3072 * 2
    oa, ob = a.forward(xd, sd_).clone(), b.forward(xd, sd_).clone()
    assert torch.equal(oa, ob)
    # one backward each: the re-made activations are the forward's, bit for bit (compared where they are consumed: the saved gelu' tile of the
    # last layer processed = layer 0 sits in the recompute engine's single slot and in slot 0 of the default engine's per-layer array)
    L.call("hs_mse_grad", oa.data_ptr(), a.dout.data_ptr(), None, oa.numel(), Fn._s())
    L.call("hs_mse_grad", ob.data_ptr(), b.dout.data_ptr(), None, ob.numel(), Fn._s())
    a.backward(a.dout)
    b.backward(b.dout)
    torch.cuda.synchronize()
    # float atomics (split-K weight gradients, LayerNorm's column sums) make two runs of ONE engine differ in the last bits too: the
    # comparison is to that noise level, far below the 1e-2 of a wrong activation
    noise = float((a.grad - b.grad).norm() / a.grad.norm())
    print(f"gradient difference with / without MLP recompute: {noise:.1e}")
    assert noise < 2e-6
    for _ in range(2):
        a.train_step(xd, sd_)
        b.train_step(xd, sd_)
    torch.cuda.synchronize()
    assert float((a.flat - b.flat).norm() / a.flat.norm()) < 1e-5
    # the C entry point refuses the option outside the 16-bit paths (bf16, and fp16 since round 5)
    assert L.load().hs_vit_workspace_bytes(C.byref(L.VitDesc(1, 65, 768, 12, 3072, 2, L.PREC_F32, 2, 0.0, None))) == 0
    assert L.load().hs_vit_workspace_bytes(C.byref(L.VitDesc(1, 65, 768, 12, 3072, 2, L.PREC_F16, 2, 0.0, None))) > 0
    assert L.load().hs_vit_workspace_bytes(C.byref(L.VitDesc(1, 65, 768, 12, 3072, 2, L.PREC_BF16, 3, 0.0, None))) == 0
