"""Round 6: the fused optimizer sweep (hs_adamw_fused), caller-owned 16-bit weight copies (hs_vit_desc.w16 / .w16t, hs_vit_cast_weights),
and the fused autograd nodes of the module API (fused.ChunkFn / PackerFn, the W16Cache, shared patches) -- the pieces behind the drop-in
training rate (bench.py `module_api_train`) and the one-volume step.  Everything goes through the C ABI (ctypes)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


def _modules(dev, seed, layers=2, train=False):
    import hsenet_amd as H
    from oracle import ref_cpu as R
    tower = H.build_vision_tower(Cfg(IMG, PATCH))
    for t in (tower.vision_tower_stage1, tower.vision_tower_stage2):
        del t.blocks[layers:]
    p1 = H.VisualPacker_3d_phi_v3(*PARGS, grid=GRID, kernel_size=KERN)
    p2 = H.VisualPacker_3d_phi_v3(*PARGS, grid=GRID, kernel_size=KERN)
    for i, m in enumerate((tower, p1, p2)):
        m.load_state_dict(R.seeded_state_dict(m, seed + i))
        m.to(dev)
        m.train(train)
    return tower, p1, p2


def _inputs(B, seed, dev):
    from oracle import ref_cpu as R
    return R.synthetic_volume((B, 1) + IMG, seed).to(dev), R.synthetic_slices(B, seed + 1).to(dev)


# ------------------------------------------------------------------------------------------------------------------------------------
# hs_adamw_fused
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("f16", [0, 1])
def test_adamw_fused_equals_masked_and_writes_the_copies(dev, f16):
    """the fused sweep's moments are BIT-equal to hs_adamw_masked's over three steps (clip active) and its parameters equal to an ulp (the
    bias corrections 1 - beta^step come from the device's powf here and from the host's there); g is zeroed behind the read; the mapped
    chunks' 16-bit copies equal a cast of the updated parameters and unmapped destinations stay untouched; the step counter is on the device."""
    from hsenet_amd import _lib as L
    torch.manual_seed(3)
    n = 64 * 1000
    p0 = torch.randn(n, device=dev)
    mask = (torch.rand(n // 64, device=dev) > 0.3).to(torch.uint8)
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    gsq, scratch = torch.zeros(1, device=dev), torch.zeros(1024, device=dev)
    state = torch.zeros(8, dtype=torch.int32, device=dev)
    dt = torch.float16 if f16 else torch.bfloat16
    w16 = torch.full((n,), 7.0, dtype=dt, device=dev)
    table = torch.zeros(n // 64, dtype=torch.int64)
    mapped = torch.zeros(n // 64, dtype=torch.bool)
    mapped[100:400] = True                                  # chunks 100..399 -> w16[64 * chunk ...] (the same position, for the comparison)
    table[mapped] = w16.data_ptr() + 128 * torch.arange(n // 64)[mapped]
    table_d = table.to(dev)
    s = torch.cuda.current_stream().cuda_stream
    for step in (1, 2, 3):
        g = torch.randn(n, device=dev) * (5.0 if step == 2 else 0.01)     # step 2 clips
        ga, gb = g.clone(), g.clone()
        L.call("hs_sumsq", ga.data_ptr(), gsq.data_ptr(), scratch.data_ptr(), n, s)
        L.call("hs_adamw_masked", pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), mask.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5,
               gsq.data_ptr(), 1.0, step, s)
        L.call("hs_adamw_fused", pb.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), mask.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5,
               gsq.data_ptr(), 1.0, 1, table_d.data_ptr(), f16, state.data_ptr(), L.ADAMW_STEP, s)
        # (the clip coefficient is derived in a one-thread prologue there and per thread here: an ulp apart under FMA contraction)
        assert rel(pb, pa) < 1e-6 and rel(mb, ma) < 1e-6 and rel(vb, va) < 1e-6, step
        if step == 1:
            assert torch.equal(ma, mb) and torch.equal(va, vb)
        assert float(gb.abs().max()) == 0.0 and torch.equal(ga, g)
        want = pb.to(dt)
        m64 = mapped.to(dev).repeat_interleave(64)
        assert torch.equal(w16[m64], want[m64]) and bool((w16[~m64] == 7.0).all())
        assert state[:2].tolist() == [step, 0]
    # a non-finite gradient norm: nothing moves, the gradient is discarded, the skip is counted and the step is not
    gb = torch.randn(n, device=dev)
    gsq.fill_(float("inf"))
    before = (pb.clone(), mb.clone(), vb.clone(), w16.clone())
    L.call("hs_adamw_fused", pb.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), mask.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5,
           gsq.data_ptr(), 1.0, 1, table_d.data_ptr(), f16, state.data_ptr(), L.ADAMW_STEP, s)
    assert all(torch.equal(a, b) for a, b in zip(before, (pb, mb, vb, w16))) and float(gb.abs().max()) == 0.0
    assert state[:2].tolist() == [3, 1]
    # the next good step is step 4 (bias corrections from the device counter), zero_grad = 0 leaves g alone, no norm = no clip
    g = torch.randn(n, device=dev) * 0.01
    ga, gb = g.clone(), g.clone()
    L.call("hs_adamw_masked", pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), mask.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5,
           None, 0.0, 4, s)
    L.call("hs_adamw_fused", pb.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), mask.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5,
           None, 0.0, 0, None, f16, state.data_ptr(), L.ADAMW_STEP, s)
    assert rel(pb, pa) < 1e-6 and torch.equal(gb, g) and state[:2].tolist() == [4, 1]
    # the pass split for overlap: ONE prologue, then sweeps over disjoint ranges in any order = the one-call pass, bit for bit
    pc, mc, vc = pb.clone(), mb.clone(), vb.clone()
    g = torch.randn(n, device=dev) * 0.01
    gb, gc = g.clone(), g.clone()
    st2 = state.clone()
    L.call("hs_sumsq", gb.data_ptr(), gsq.data_ptr(), scratch.data_ptr(), n, s)
    wb, wc = w16.clone(), w16.clone()
    tb = table.clone()
    tb[mapped] = wb.data_ptr() + 128 * torch.arange(n // 64)[mapped]
    tc = table.clone()
    tc[mapped] = wc.data_ptr() + 128 * torch.arange(n // 64)[mapped]
    tb, tc = tb.to(dev), tc.to(dev)
    common = (1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5, gsq.data_ptr(), 1.0, 1)
    L.call("hs_adamw_fused", pb.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), mask.data_ptr(), n, *common, tb.data_ptr(), f16,
           state.data_ptr(), L.ADAMW_STEP, s)
    L.call("hs_adamw_fused", None, None, None, None, None, 0, *common, None, f16, st2.data_ptr(), L.ADAMW_PREPARE, s)
    for a, b in ((64 * 600, n), (0, 64 * 200), (64 * 200, 64 * 600)):
        L.call("hs_adamw_fused", pc.data_ptr() + 4 * a, gc.data_ptr() + 4 * a, mc.data_ptr() + 4 * a, vc.data_ptr() + 4 * a, mask.data_ptr() + a // 64,
               b - a, *common, tc.data_ptr() + 8 * (a // 64), f16, st2.data_ptr(), L.ADAMW_SWEEP, s)
    assert torch.equal(pb, pc) and torch.equal(mb, mc) and torch.equal(vb, vc) and torch.equal(wb, wc) and float(gc.abs().max()) == 0.0
    assert st2[:2].tolist() == state[:2].tolist() == [5, 1]


# ------------------------------------------------------------------------------------------------------------------------------------
# caller-owned weight copies
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_vit_cast_weights_and_external_copies_give_the_same_bits(dev, prec):
    """hs_vit_cast_weights: straight copies = cast(W), transposed = cast(W)^T, from fp32 and from the 16-bit set alike; the block driver
    with caller-owned copies (desc.w16 / .w16t) produces bit-identical outputs and (atomics aside) gradients to the one that casts per call."""
    from hsenet_amd import _lib as L, functional as Fn
    P = {"bf16": L.PREC_BF16, "f16": L.PREC_F16}[prec]
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[prec]
    torch.manual_seed(5)
    B, N, d, H, mlp, Ln = 2, 65, 128, 2, 192, 2
    shapes = [(mlp, d), (mlp,), (d, mlp), (d,), (d,), (d,), (d, d), (d,), (3 * d, d), (d,), (d,)]
    params = [torch.randn(*s, device=dev) * (0.05 if len(s) == 2 else 0.1) for _ in range(Ln) for s in shapes]
    normp = [torch.ones(d, device=dev), torch.zeros(d, device=dev)]
    lp, npar = Fn._ptr_array(params), Fn._ptr_array(normp)
    desc = L.VitDesc(B, N, d, H, mlp, Ln, P, 1, 0.0, None)
    lib, s = L.load(), torch.cuda.current_stream().cuda_stream
    n = lib.hs_vit_weights_elems(C.byref(desc))
    per = 4 * d * d + 2 * d * mlp
    assert n == Ln * per
    w16, w16t, w16t2 = (torch.zeros(n, dtype=dt, device=dev) for _ in range(3))
    L.call("hs_vit_cast_weights", C.byref(desc), lp, w16.data_ptr(), w16t.data_ptr(), s)
    L.call("hs_vit_cast_weights", C.byref(desc), None, w16.data_ptr(), w16t2.data_ptr(), s)
    assert torch.equal(w16t, w16t2)
    for l in range(Ln):
        o = l * per
        for idx, (r, c) in ((8, (3 * d, d)), (6, (d, d)), (0, (mlp, d)), (2, (d, mlp))):
            W = params[l * 11 + idx].to(dt)
            assert torch.equal(w16[o:o + r * c].view(r, c), W) and torch.equal(w16t[o:o + r * c].view(c, r), W.t())
            o += r * c
    x0, dy = torch.randn(B, N, d, device=dev), torch.randn(B, N, d, device=dev)

    def run(ext):
        dsc = L.VitDesc(B, N, d, H, mlp, Ln, P, 1, 0.0, None)
        if ext:
            dsc.w16, dsc.w16t = w16.data_ptr(), w16t.data_ptr()
        nb = lib.hs_vit_workspace_bytes(C.byref(dsc))
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
        wp = ws.data_ptr() + (-ws.data_ptr()) % 256
        y, hid, dx0 = torch.empty_like(x0), torch.empty(Ln, B, N, d, device=dev), torch.empty_like(x0)
        grads = [torch.zeros_like(p) for p in params + normp]
        L.call("hs_vit_blocks_fwd", C.byref(dsc), x0.data_ptr(), lp, npar, y.data_ptr(), hid.data_ptr(), wp, nb, s)
        L.call("hs_vit_blocks_bwd", C.byref(dsc), x0.data_ptr(), lp, npar, dy.data_ptr(), dx0.data_ptr(), Fn._ptr_array(grads[:-2]),
               Fn._ptr_array(grads[-2:]), hid.data_ptr(), wp, nb, Ln, 0, s)
        torch.cuda.synchronize()
        return nb, y, dx0, grads
    nb0, y0, dx00, g0 = run(False)
    nb1, y1, dx01, g1 = run(True)
    assert nb1 == nb0 - 2 * (2 * n + (-(2 * n)) % 256) or nb1 < nb0
    assert torch.equal(y0, y1) and torch.equal(dx00, dx01)
    assert max(rel(a, b) for a, b in zip(g1, g0)) < 1e-5


def test_engine_weight_copies_follow_every_kind_of_parameter_change(dev):
    """the engine keeps its encoders' 16-bit weight copies OUTSIDE the step: written by the optimizer sweep, re-made when torch writes a
    parameter (load_state_dict, an in-place op on a parameter view, a copy into the flat buffer).  After any of these the next forward must
    equal that of a fresh engine built on the same values."""
    from hsenet_amd.engine import DualEngine
    x, s2d = _inputs(2, 40, dev)

    def fresh_out(state):
        t, a, b = _modules(dev, 30)
        for m, sd in zip((t, a, b), state):
            m.load_state_dict(sd)
        return DualEngine(t, a, b, 2, lr=1e-3).forward(x, s2d).clone()
    tower, p1, p2 = _modules(dev, 30)
    eng = DualEngine(tower, p1, p2, 2, lr=1e-2, keep_grads=False)
    snap = lambda: [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (tower, p1, p2)]
    for _ in range(2):
        eng.train_step(x, s2d)                                  # the sweep writes w16, after_step() re-makes w16t
    assert eng.step_no == 2 and eng.skipped_steps == 0
    assert torch.equal(eng.forward(x, s2d), fresh_out(snap()))
    u = eng.u1
    W = tower.vision_tower_stage1.blocks[1].mlp.linear1.weight
    per = u.w_elems // u.Ln
    got = u.w16[per + 4 * 768 * 768: per + 4 * 768 * 768 + W.numel()].view(W.shape)
    assert torch.equal(got, W.detach().to(torch.bfloat16))      # the optimizer's copy IS the cast of the updated master
    with torch.no_grad():
        tower.vision_tower_stage2.blocks[0].attn.qkv.weight[:768].mul_(1.5)      # in place through a parameter view (bench.py --attn-logit-std)
    assert torch.equal(eng.forward(x, s2d), fresh_out(snap()))
    sd = snap()
    sd[0]["vision_tower_stage1.blocks.0.attn.out_proj.weight"].mul_(0.5)
    tower.load_state_dict(sd[0])                                                # load_state_dict after adoption
    assert torch.equal(eng.forward(x, s2d), fresh_out(sd))
    saved = eng.flat.clone()
    eng.train_step(x, s2d)
    eng.flat.copy_(saved)                                                       # restoring through the flat buffer
    assert torch.equal(eng.forward(x, s2d), fresh_out(sd))
    # and the BACKWARD after such a change multiplies with fresh transposed copies: same gradient as a fresh engine on these values
    eng.train_step(x, s2d, optimizer=False)
    t2, a2, b2 = _modules(dev, 30)
    for m, st in zip((t2, a2, b2), sd):
        m.load_state_dict(st)
    e2 = DualEngine(t2, a2, b2, 2, lr=1e-2)
    e2.train_step(x, s2d, optimizer=False)
    assert rel(eng.grad, e2.grad) < 1e-5


def test_fused_optimizer_sweep_equals_keep_grads_path(dev):
    """keep_grads=False (the default outside the test-suite: gradient zeroed inside the optimizer sweep, weight copies written by it) and
    keep_grads=True (gradient left in place, cleared at the next backward) walk the same trajectory -- as close as two runs of ONE of them
    are to each other (the weight gradients accumulate with float atomics, and Adam's first steps are sign-like); bf16 and fp16, where a
    poisoned step is skipped, counted, and does not advance the step number."""
    from hsenet_amd.engine import DualEngine
    x, s2d = _inputs(2, 50, dev)
    start = torch.cat([p.detach().flatten() for m in _modules(dev, 60) for p in m.parameters()])
    for prec in ("bf16", "f16"):
        upd = []
        for keep in (True, True, False):
            t, a, b = _modules(dev, 60)
            eng = DualEngine(t, a, b, 2, lr=1e-3, precision=prec, keep_grads=keep, two_streams=False)
            assert eng.keep_grads is keep
            for _ in range(3):
                eng.train_step(x, s2d)
            assert (float(eng.grad.abs().max()) == 0.0) == (not keep)
            now = torch.cat([p.detach().flatten() for m in (t, a, b) for p in m.parameters()])
            upd.append(now - start)
            if prec == "f16" and not keep:
                before = eng.flat.clone()
                eng.forward(x, s2d)
                eng.dout.fill_(float("inf"))
                eng.backward(eng.dout)
                eng.optimizer_step()
                assert eng.skipped_steps == 1 and eng.step_no == 3 and torch.equal(eng.flat, before)
                assert float(eng.grad.abs().max()) == 0.0            # the overflowed gradient was discarded by the sweep
                eng.train_step(x, s2d)
                assert eng.step_no == 4 and eng.skipped_steps == 1 and bool(torch.isfinite(eng.flat).all())
        noise, diff = rel(upd[1], upd[0]), rel(upd[2], upd[0])
        assert diff < max(5 * noise, 5e-3), (prec, noise, diff)      # (a real fault -- stale copies, a gradient not zeroed -- shows as >= 1e-1)


def test_overlapped_optimizer_pass_walks_the_same_trajectory(dev):
    """overlap_optimizer=True (an option, off by default: it buys +0.5 / +1 % at 1 / 3 volumes per GPU): the optimizer pass sweeps the flat buffer chunk by chunk on a third
    stream, the next forward's chunks wait for their events only, packers and backward for the whole pass.  Same trajectory as the
    synchronous pass over five steps with two streams on (bf16 and fp16), a forward / state_dict read in between, an in-place weight
    edit while a pass may be in flight, and the plan partitions the flat buffer."""
    from hsenet_amd.engine import DualEngine, FlatEncoder
    import hsenet_amd as H
    x, s2d = _inputs(2, 140, dev)
    start = torch.cat([p.detach().flatten() for m in _modules(dev, 150) for p in m.parameters()])
    for prec in ("bf16", "f16"):
        upd = []
        for overlap in (False, False, True):
            t, a, b = _modules(dev, 150)
            eng = DualEngine(t, a, b, 2, lr=1e-3, precision=prec, keep_grads=False, overlap_optimizer=overlap, two_streams=True)
            assert eng.overlap_optimizer is overlap
            if overlap:
                plan = eng._opt_plan()
                assert sorted((a_, b_) for a_, b_, _, _ in plan)[0][0] == 0 and sum(b_ - a_ for a_, b_, _, _ in plan) == eng.params.total
                assert [k for _, _, u, k in plan if u is eng.u1] == [0, 1] and [u is None for _, _, u, _ in plan][-1]
            for i in range(5):
                eng.train_step(x, s2d)
                if i == 1:
                    y = eng.forward(x, s2d).clone()                  # an extra forward between steps (waits per chunk, packers for the pass)
                    eng.wait_optimizer()
                    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(eng.flat).all())
                if i == 2:
                    with torch.no_grad():
                        t.vision_tower_stage1.blocks[1].attn.qkv.weight.mul_(1.0)      # a version bump while the pass may be in flight
            eng.wait_optimizer()
            torch.cuda.synchronize()
            assert eng.step_no == 5 and eng.skipped_steps == 0 and float(eng.grad.abs().max()) == 0.0
            now = torch.cat([p.detach().flatten() for m in (t, a, b) for p in m.parameters()])
            upd.append(now - start)
        noise, diff = rel(upd[1], upd[0]), rel(upd[2], upd[0])
        assert diff < max(5 * noise, 5e-3), (prec, noise, diff)      # (a real fault -- stale copies, a gradient not zeroed -- shows as >= 1e-1)
    # one encoder (configs[1]) and the 16-bit copies after an overlapped pass: what the next forward multiplies with IS the updated master
    m = H.ViT_stage1(in_channels=1, img_size=IMG, patch_size=PATCH, num_layers=2, pos_embed="perceptron", classification=True).to(dev)
    fe = FlatEncoder(m, 2, lr=1e-2, keep_grads=False, overlap_optimizer=True)
    for _ in range(2):
        fe.train_step(x)
    fe.wait_optimizer()
    torch.cuda.synchronize()
    W = m.blocks[1].mlp.linear2.weight
    per = fe.unit.w_elems // fe.unit.Ln
    o = per + 4 * 768 * 768 + 3072 * 768
    assert torch.equal(fe.unit.w16[o:o + W.numel()].view(W.shape), W.detach().to(torch.bfloat16))
    assert torch.equal(fe.unit.w16t[o:o + W.numel()].view(W.shape[1], W.shape[0]), W.detach().to(torch.bfloat16).t())


# ------------------------------------------------------------------------------------------------------------------------------------
# module API: fused nodes
# ------------------------------------------------------------------------------------------------------------------------------------
def _module_step(mods, x, s2d, scale=1.0):
    import hsenet_amd as H
    for m in mods:
        m.zero_grad(set_to_none=True)
    out = H.encode_images(*mods, x, s2d)
    (out.square().mean() * scale).backward()
    torch.cuda.synchronize()
    return out.detach(), {f"{i}.{n}": p.grad.detach() / scale for i, m in enumerate(mods) for n, p in m.named_parameters()}


@pytest.mark.parametrize("prec,tol_o,tol_g", [("bf16", 7e-3, 4e-2), ("f16", 8e-4, 6e-3)])
def test_fused_module_nodes_equal_the_per_operator_functions(dev, prec, tol_o, tol_g, monkeypatch):
    """encode_images through the fused autograd nodes (fused.ChunkFn: both encoders, embedding + blocks, cached weight copies, the 2E3
    chain on a side stream inside each node; PackerFn reading the [:, 1:] view in place) against the same modules on one Function per reference operator (HS_MODULE_UNFUSED=1): outputs and every one of the 316
    parameter gradients agree to the operand format's rounding (the two forms round intermediate tensors at different places), and both
    agree with the exact-f32 mode."""
    import hsenet_amd as H
    x, s2d = _inputs(2, 70, dev)
    mods = _modules(dev, 80)
    ls = 1024.0 if prec == "f16" else 1.0
    H.set_precision("f32")
    try:
        o32, g32 = _module_step(mods, x, s2d)
        H.set_precision(prec)
        of, gf = _module_step(mods, x, s2d, ls)
        monkeypatch.setenv("HS_MODULE_UNFUSED", "1")
        ou, gu = _module_step(mods, x, s2d, ls)
        monkeypatch.delenv("HS_MODULE_UNFUSED")
    finally:
        H.set_precision("bf16")
    assert set(gf) == set(gu) == set(g32) and len(gf) == 2 * 2 * 11 + 2 * 2 + 4 + 4 + 12 + 2 * 14
    assert rel(of, ou) < tol_o and rel(of, o32) < 2 * tol_o and rel(ou, o32) < 2 * tol_o
    gn = lambda g: torch.cat([v.flatten() for v in g.values()])
    assert rel(gn(gf), gn(gu)) < tol_g and rel(gn(gf), gn(g32)) < tol_g
    worst = max(rel(gf[k], g32[k]) for k in gf if float(g32[k].norm()) > 1e-3 * float(gn(g32).norm()))
    worst_u = max(rel(gu[k], g32[k]) for k in gu if float(g32[k].norm()) > 1e-3 * float(gn(g32).norm()))
    assert worst < max(2.5 * worst_u, 2 * tol_g), (worst, worst_u)


def test_module_weight_cache_follows_the_parameters(dev):
    """the module API casts the block weights when they CHANGE (torch's version counters), not per call: an optimizer step, an in-place
    edit or load_state_dict between two forwards must show in the second one; a backward after such an edit is refused by torch itself."""
    import hsenet_amd as H
    x, s2d = _inputs(2, 90, dev)
    tower, p1, p2 = _modules(dev, 100)
    fresh, f1, f2 = _modules(dev, 100)

    def same_as_fresh():
        for a, b in zip((tower, p1, p2), (fresh, f1, f2)):
            b.load_state_dict(a.state_dict())
        for m in fresh.modules():
            m.__dict__.pop("_w16cache", None)
        with torch.no_grad():
            return torch.equal(H.encode_images(tower, p1, p2, x, s2d), H.encode_images(fresh, f1, f2, x, s2d))
    with torch.no_grad():
        y0 = H.encode_images(tower, p1, p2, x, s2d).clone()
        cache = tower.vision_tower_stage1.__dict__["_w16cache"]
        key0 = cache._sets[0][2]
        assert torch.equal(H.encode_images(tower, p1, p2, x, s2d), y0) and cache._sets[0][2] == key0      # second call: no re-cast
        tower.vision_tower_stage1.blocks[0].mlp.linear2.weight.mul_(1.25)
        y1 = H.encode_images(tower, p1, p2, x, s2d)
        assert cache._sets[0][2] != key0 and not torch.equal(y1, y0)
    assert same_as_fresh()
    opt = torch.optim.AdamW([p for m in (tower, p1, p2) for p in m.parameters()], lr=1e-2, fused=True)
    out = H.encode_images(tower, p1, p2, x, s2d)
    out.square().mean().backward()
    opt.step()
    assert same_as_fresh()
    # a write torch does not record (an alias with its own version counter): the documented escape hatch
    tower.vision_tower_stage2.blocks[0].mlp.linear1.weight.data.mul_(0.5)
    assert not same_as_fresh()                 # stale copies (`.data` writes need invalidate_weight_copies())
    H.invalidate_weight_copies()
    assert same_as_fresh()
    out = H.encode_images(tower, p1, p2, x, s2d)
    with torch.no_grad():
        tower.vision_tower_stage2.blocks[1].attn.qkv.weight.add_(0.01)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.square().mean().backward()


def test_module_train_step_matches_the_engine_step(dev):
    """one train step written the reference's way (module API + autograd + torch.optim.AdamW with HF Trainer's groups + clip_grad_norm_)
    lands where engine.DualEngine.train_step lands: same loss, same parameters after the step (eval(): no dropout draws)."""
    import hsenet_amd as H
    from bench_module_api import hf_param_groups
    from hsenet_amd.engine import DualEngine
    x, s2d = _inputs(2, 110, dev)
    mods = _modules(dev, 120)
    ref = _modules(dev, 120)
    params = [p for m in mods for p in m.parameters()]
    opt = torch.optim.AdamW(hf_param_groups(mods, 0.1), lr=1e-3, fused=True)
    for _ in range(2):
        out = H.encode_images(*mods, x, s2d)
        loss = out.square().mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
    eng = DualEngine(*ref, 2, lr=1e-3, weight_decay=0.1, keep_grads=False)
    for _ in range(2):
        eng.train_step(x, s2d)
    assert abs(float(eng.loss) - float(loss)) < 2e-3 * abs(float(loss))
    mine = torch.cat([p.detach().flatten() for m in mods for p in m.parameters()])
    theirs = torch.cat([p.detach().flatten() for m in ref for p in m.parameters()])
    start = torch.cat([p.detach().flatten() for m in _modules(dev, 120) for p in m.parameters()])
    assert rel(mine - start, theirs - start) < 3e-2          # the UPDATES agree (Adam's sign-like first steps amplify bf16 gradient noise)
    assert rel(mine, theirs) < 5e-4


def test_packer_fn_reads_the_patch_view_in_place(dev):
    """PackerFn on the reference's `features[:, 1:]` view: same output and input gradient as on a contiguous copy; the gradient arrives at
    the full [B, T + 1, d] tensor with a zero cls row."""
    import hsenet_amd as H
    torch.manual_seed(9)
    _, p1, _ = _modules(dev, 130)
    T = GRID[0] * GRID[1] * GRID[2]
    full = torch.randn(3, T + 1, 768, device=dev, requires_grad=True)
    o1 = p1(full[:, 1:])
    o1.square().mean().backward()
    g1 = full.grad.clone()
    full.grad = None
    p1.zero_grad(set_to_none=True)
    o2 = p1(full[:, 1:].contiguous())
    o2.square().mean().backward()
    assert torch.equal(o1, o2) and float(g1[:, 0].abs().max()) == 0.0
    assert rel(g1, full.grad) < 1e-5
    odd = torch.randn(3, T, 2 * 768, device=dev)[:, :, :768]                  # not a row-offset view: packed first
    assert torch.equal(p1(odd), p1(odd.contiguous()))


@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("f16", 1.5e-3)])
def test_default_conv_embedding_on_the_fused_paths(dev, prec, tol):
    """pos_embed='conv' is the constructors' DEFAULT (vit.py:237,375).  On one-channel volumes -- every call site of the reference -- the
    Conv3d weight [hidden, 1, p1, p2, p3] IS the perceptron's [hidden, p1*p2*p3] matrix in memory, so the fused node chain and the
    engines take it as it stands (state-dict keys `patch_embeddings.weight / .bias`); forward and gradients against the oracle's
    F.conv3d form."""
    import hsenet_amd as H
    from hsenet_amd.engine import FlatEncoder
    from oracle import ref_cpu as R
    kw = dict(in_channels=1, img_size=IMG, patch_size=PATCH, num_layers=2, pos_embed="conv", classification=True)
    ref = R.ViT_stage2(**kw).eval()
    sd = R.seeded_state_dict(ref, 4300)
    ref.load_state_dict(sd)
    x, s2d = R.synthetic_volume((2, 1) + IMG, 4301), R.synthetic_slices(2, 4302)
    yr, _ = ref(x, s2d)
    (yr * yr).mean().backward()
    want = {n: p.grad for n, p in ref.named_parameters()}
    ls = 1024.0 if prec == "f16" else 1.0
    H.set_precision(prec)
    try:
        m = H.ViT_stage2(**kw).eval()
        m.load_state_dict(sd)
        m = m.to(dev)
        assert list(m.state_dict())[:4] == list(ref.state_dict())[:4] and "patch_embedding.patch_embeddings.weight" in m.state_dict()
        assert m._fused_ok(H.encoder._prec(m), x)                       # the fused chain, not the per-operator fallback
        y, hidden = m(x.to(dev), s2d.to(dev))
        ((y * y).mean() * ls).backward()
    finally:
        H.set_precision("bf16")
    assert rel(y.detach(), yr.detach()) < tol and len(hidden) == 2
    gtol = 4e-2 if prec == "bf16" else 8e-3
    for n in ("patch_embedding.patch_embeddings.weight", "patch_embedding.patch_embeddings.bias", "patch_embedding.position_embeddings",
              "blocks.0.attn.qkv.weight", "slice_guided_attention.Wq.weight"):
        assert rel(dict(m.named_parameters())[n].grad / ls, want[n]) < gtol, n
    m.zero_grad(set_to_none=True)
    eng = FlatEncoder(m, 2, lr=1e-3, precision=prec, keep_grads=True)
    eng.train_step(x.to(dev), s2d.to(dev), optimizer=False)
    got = dict(m.named_parameters())
    for n in ("patch_embedding.patch_embeddings.weight", "patch_embedding.patch_embeddings.bias", "blocks.1.mlp.linear1.weight"):
        assert rel(got[n].grad / eng.loss_scale, want[n]) < gtol, n


@pytest.mark.parametrize("prec,tol,gtol", [("f32", 1e-5, 1e-4), ("bf16", 1e-2, 4e-2), ("f16", 1.5e-3, 8e-3)])
def test_qkv_bias_true_trains_on_every_path(dev, prec, tol, gtol):
    """qkv_bias=True (accepted by the reference's constructors, vit.py:385; no shipped call site passes it): the biased projection with its
    q third pre-scaled, and the bias gradient as the column sums of dqkv -- module API (fused chain at 16 bits, per-operator Functions at
    f32) and the engine, forward and every gradient against the oracle."""
    import hsenet_amd as H
    from hsenet_amd.engine import FlatEncoder
    from oracle import ref_cpu as R
    kw = dict(in_channels=1, img_size=IMG, patch_size=PATCH, num_layers=2, pos_embed="perceptron", classification=True, qkv_bias=True)
    ref = R.ViT_stage1(**kw).eval()
    sd = R.seeded_state_dict(ref, 4400)
    ref.load_state_dict(sd)
    x = R.synthetic_volume((2, 1) + IMG, 4401)
    yr, _ = ref(x)
    (yr * yr).mean().backward()
    want = {n: p.grad for n, p in ref.named_parameters()}
    assert float(want["blocks.0.attn.qkv.bias"].norm()) > 0
    ls = 1024.0 if prec == "f16" else 1.0
    H.set_precision(prec)
    try:
        m = H.ViT_stage1(**kw).eval()
        m.load_state_dict(sd)
        m = m.to(dev)
        y, _ = m(x.to(dev))
        ((y * y).mean() * ls).backward()
    finally:
        H.set_precision("bf16")
    assert rel(y.detach(), yr.detach()) < tol
    got = dict(m.named_parameters())
    assert set(got) == set(want)
    for n in want:
        if float(want[n].norm()) > 1e-3 * max(float(w.norm()) for w in want.values()):
            assert rel(got[n].grad / ls, want[n]) < gtol, n
    if prec == "f32":
        return
    m.zero_grad(set_to_none=True)
    eng = FlatEncoder(m, 2, lr=1e-3, precision=prec, keep_grads=True)
    eng.train_step(x.to(dev), optimizer=False)
    for n in ("blocks.0.attn.qkv.bias", "blocks.1.attn.qkv.bias", "blocks.0.attn.qkv.weight", "blocks.1.mlp.linear2.bias"):
        assert rel(got[n].grad / eng.loss_scale, want[n]) < gtol, n
    eng.train_step(x.to(dev))                                     # and a full step moves the biases
    assert not torch.equal(got["blocks.0.attn.qkv.bias"].detach().cpu(), sd["blocks.0.attn.qkv.bias"])


def test_two_live_graphs_and_gradient_accumulation_through_the_fused_chain(dev):
    """two forward calls before one backward (micro-batches of a gradient-accumulation step, or two towers' calls in one loss): each call
    owns its workspaces and its chunk chain -- the residual-stream gradient that travels inside a chain's workspace cannot be confused
    with the other chain's -- and the weight copies are shared read-only.  Gradients of loss(x1) + loss(x2) in one backward equal the sum
    of two separate backward passes, and equal accumulation into existing .grad tensors."""
    import hsenet_amd as H
    mods = _modules(dev, 160)
    x1, s1 = _inputs(2, 170, dev)
    x2, s2 = _inputs(2, 180, dev)
    params = [p for m in mods for p in m.parameters()]

    def loss(x, s):
        return H.encode_images(*mods, x, s).square().mean()
    for m in mods:
        m.zero_grad(set_to_none=True)
    (loss(x1, s1) + loss(x2, s2)).backward()                         # two live graphs, one backward
    both = torch.cat([p.grad.flatten() for p in params]).clone()
    for m in mods:
        m.zero_grad(set_to_none=True)
    loss(x1, s1).backward()
    g1 = torch.cat([p.grad.flatten() for p in params]).clone()
    loss(x2, s2).backward()                                          # accumulates into the existing .grad (AccumulateGrad adds in place)
    acc = torch.cat([p.grad.flatten() for p in params]).clone()
    for m in mods:
        m.zero_grad(set_to_none=True)
    loss(x2, s2).backward()
    g2 = torch.cat([p.grad.flatten() for p in params]).clone()
    assert rel(both, g1 + g2) < 1e-5 and rel(acc, g1 + g2) < 1e-5 and rel(g1, g2) > 1e-2
    # retain_graph: a second backward through the same chain re-walks the workspaces from the top
    for m in mods:
        m.zero_grad(set_to_none=True)
    l = loss(x1, s1)
    l.backward(retain_graph=True)
    l.backward()
    twice = torch.cat([p.grad.flatten() for p in params])
    assert rel(twice, 2 * g1) < 1e-5
