"""Per-kernel parity tests (GPU): each HIP kernel, called through the C ABI, against a plain torch fp32/fp64
statement of the same op on the same seeded inputs.  Tolerances are written next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(shape, seed, dev, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy(g.standard_normal(shape, dtype=np.float32) * scale).to(dev)


GEMM_SHAPES = [(128, 128, 64), (300, 200, 192), (4097, 768, 768), (37, 2304, 128), (513, 8, 320), (64, 136, 72)]


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_gemm_layouts(dev, layout, M, N, K, prec):
    from hsenet_amd import functional as Fn, _lib as L
    if layout == 2 and M % 8:
        pytest.skip("TN needs M % 8 == 0 (16-B chunks along the contiguous axis)")
    if layout != 0 and N % 8:
        pytest.skip("NN/TN need N % 8 == 0")
    P = L.PREC_BF16 if prec == "bf16" else L.PREC_F32
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    a = rnd((M, K), 1, dev).to(dt)
    b = rnd((N, K), 2, dev).to(dt)
    ref = a.double() @ b.double().t()
    A = a if layout != 2 else a.t().contiguous()
    Bm = b if layout == 0 else b.t().contiguous()
    out = torch.full((M, N), float("nan"), device=dev)
    Fn.gemm(A, Bm, layout, prec=P, out_f32=out)
    torch.cuda.synchronize()
    # same (rounded) inputs on both sides, fp32 accumulation: only summation order differs -> 2e-6 * sqrt(K) headroom
    assert rel(out, ref) < 1e-5, rel(out, ref)


def test_gemm_epilogues(dev):
    from hsenet_amd import functional as Fn, _lib as L
    M, N, K = 333, 256, 128
    a, b = rnd((M, K), 3, dev).bfloat16(), rnd((N, K), 4, dev).bfloat16()
    bias, resid = rnd((N,), 5, dev), rnd((M, N), 6, dev)
    acc = a.double() @ b.double().t()
    # bias + gelu, saving the bf16 pre-activation
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    z16 = torch.empty_like(o16)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16, bias=bias, aux_out=z16, flags=L.EPI_BIAS | L.EPI_GELU)
    z = (acc + bias.double()).float()
    assert rel(z16.float(), z) < 4e-3                      # bf16 rounding of the output: 2^-9
    assert rel(o16.float(), F.gelu(z16.float())) < 4e-3    # gelu is evaluated on the rounded pre-activation
    # bias + residual -> fp32
    o32 = torch.empty(M, N, device=dev)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_f32=o32, bias=bias, resid=resid, flags=L.EPI_BIAS | L.EPI_RESID)
    assert rel(o32, acc + bias.double() + resid.double()) < 1e-5
    # gelu backward epilogue
    zz = rnd((M, N), 7, dev).bfloat16()
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_f32=o32, aux=zz, flags=L.EPI_GELU_BWD)
    zr = zz.float().double().requires_grad_()
    (g,) = torch.autograd.grad(F.gelu(zr).sum(), zr)
    assert rel(o32, acc * g) < 1e-4
    # forward keeps gelu'(z) instead of z; backward multiplies by it (HS_EPI_SAVE_DGELU)
    g16 = torch.empty_like(o16)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16, bias=bias, aux_out=g16, flags=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_DGELU)
    zq = z16.float().double().requires_grad_()
    (gq,) = torch.autograd.grad(F.gelu(zq).sum(), zq)
    assert rel(g16.float(), gq) < 4e-3 and rel(o16.float(), F.gelu(z16.float())) < 4e-3
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_f32=o32, aux=g16, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU)
    assert rel(o32, acc * g16.double()) < 1e-5
    # fused column sums of the stored values (bias gradient of a dgrad), on top of an existing value
    cs = torch.ones(N, device=dev)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16, colsum_out=cs)
    assert rel(cs, 1.0 + acc.sum(0)) < 1e-4
    # split-K atomic accumulate on top of an existing value, alpha
    Mo = 336
    base = rnd((Mo, N), 8, dev)
    a2 = rnd((1000, Mo), 9, dev).bfloat16()                         # stored [K=1000][M'=336]
    b2 = rnd((1000, N), 10, dev).bfloat16()                         # stored [K=1000][N]
    o = base[:Mo].clone()
    Fn.gemm(a2, b2, L.GEMM_TN, prec=L.PREC_BF16, out_f32=o, flags=L.EPI_ATOMIC, split_k=5, alpha=0.5)
    assert rel(o, base[:Mo].double() + 0.5 * (a2.double().t() @ b2.double())) < 1e-5


def test_gemm_batched_f32(dev):
    from hsenet_amd import functional as Fn, _lib as L
    a, b = rnd((3, 70, 36), 11, dev), rnd((3, 32, 36), 12, dev)
    out = torch.empty(3, 70, 32, device=dev)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F32, out_f32=out, alpha=0.25)
    assert rel(out, 0.25 * a.double() @ b.double().transpose(1, 2)) < 1e-6


@pytest.mark.parametrize("rows,split", [(4097, 1), (4097, 3), (520, 2), (64, 1)])
def test_gemm_group_equals_separate_wgrads(dev, rows, split):
    """hs_gemm_group (the four weight-gradient GEMMs of a layer in one launch) against fp32 matmuls and, unsplit, bit-for-bit
    against four hs_gemm calls (one atomic add per element onto zero: no order dependence)."""
    from hsenet_amd import functional as Fn, _lib as L
    shapes = [(768, 3072), (3072, 768), (2304, 768), (768, 768)]
    dys = [rnd((rows, m), 11 + i, dev).bfloat16() for i, (m, n) in enumerate(shapes)]
    xs = [rnd((rows, n), 21 + i, dev).bfloat16() for i, (m, n) in enumerate(shapes)]
    outs = [torch.zeros(m, n, device=dev) for m, n in shapes]
    Fn.wgrad_group(dys, xs, outs, split)
    for dy, x, o in zip(dys, xs, outs):
        assert rel(o, dy.float().t() @ x.float()) < 2e-3
        if split == 1:
            sep = torch.zeros_like(o)
            Fn.gemm(dy, x, L.GEMM_TN, prec=L.PREC_BF16, out_f32=sep, flags=L.EPI_ATOMIC, split_k=1)
            assert torch.equal(o, sep)
    with pytest.raises(L.HsError):
        Fn.wgrad_group(dys * 2, xs * 2, outs * 2)


def sdpa_ref(qkv, H):
    B, N, D3 = qkv.shape
    q, k, v = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    p = s.softmax(-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("N", [1, 9, 64, 65, 200, 513, 2049])
@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_attn_fwd(dev, N, prec):
    from hsenet_amd import functional as Fn, _lib as L
    B, H = 2, 3
    P = L.PREC_BF16 if prec == "bf16" else L.PREC_F32
    qkv = rnd((B, N, 3 * H * 64), 20 + N, dev, 1.5)
    qkv = qkv.bfloat16() if prec == "bf16" else qkv
    o, lse = Fn.attn_fwd(qkv, H, P)
    ro, rl = sdpa_ref(qkv, H)
    # f32: exact-f32 MFMA + fp32 softmax -> 1e-5.  bf16: P is rounded to bf16 before PV (2^-9) and O to bf16 -> 6e-3;
    # without HS_ATTN_Q_PRESCALED the bf16 kernel folds dh^-0.5*log2(e) into its q fragments (a second 2^-9 rounding of
    # q) that moves every score, and with it lse, by up to ~2^-9 sum|q_i k_i| dh^-0.5 (~1e-2 at these magnitudes);
    # test_attn_fwd_q_prescaled pins lse to 1e-5 on the layout the ViT driver actually uses
    assert rel(o.float(), ro) < (1e-5 if prec == "f32" else 6e-3), rel(o.float(), ro)
    if prec == "f32":
        assert rel(lse, rl) < 1e-5, rel(lse, rl)
    else:
        assert (lse.double() - rl).abs().max() < 3e-2, (lse.double() - rl).abs().max()


@pytest.mark.parametrize("N", [1, 65, 513, 2049])
def test_attn_fwd_q_prescaled(dev, N):
    """the layout the ViT driver uses: q third pre-multiplied by dh^-0.5*log2(e) by the qkv GEMM (one rounding)."""
    from hsenet_amd import functional as Fn, _lib as L
    B, H, D = 2, 3, 192
    raw = rnd((B, N, 3 * D), 60 + N, dev, 1.5)
    pre = raw.clone()
    pre[..., :D] *= L.Q_PRESCALE
    pre = pre.bfloat16()
    o, lse = Fn.attn_fwd(pre, H, L.PREC_BF16, q_prescaled=True)
    back = pre.double()
    back[..., :D] /= L.Q_PRESCALE                      # the q this tensor encodes, exactly
    ro, rl = sdpa_ref(back, H)
    assert rel(o.float(), ro) < 6e-3, rel(o.float(), ro)
    assert rel(lse, rl) < 1e-5, rel(lse, rl)


def test_attn_fwd_large_scores(dev):
    """rows that leave the optimistic kernel's f32 window (both sides) must come out of the guarded exact kernel, whose
    online-softmax rescale path is forced by a key that dominates from a late tile (cdna guide rule 26)."""
    from hsenet_amd import functional as Fn, _lib as L
    B, H, N = 1, 1, 300
    qkv = rnd((B, N, 192), 31, dev, 0.5)
    qkv[0, 250, 64:128] = qkv[0, 7, 0:64] * 40.0          # key 250 aligned with query 7, in the 4th key tile: +80 nats
    u = torch.nn.functional.normalize(rnd((64,), 32, dev), dim=0)
    qkv[0, :, 64:128] += 8.0 * u                          # every key has a common component ...
    qkv[0, 9, 0:64] = -60.0 * u                           # ... that query 9 opposes: all its scores ~ -60 nats
    qkv[0, 11, 0:64] = 30.0 * u                           # query 11: all scores ~ +30 nats, inside the window
    for P, dt, tol in ((L.PREC_F32, torch.float32, 1e-5), (L.PREC_BF16, torch.bfloat16, 8e-3)):
        x = qkv.to(dt)
        o, lse = Fn.attn_fwd(x, H, P)
        ro, rl = sdpa_ref(x, H)
        assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
        assert rel(o.float(), ro) < tol, rel(o.float(), ro)
        for row in (7, 9, 11, 200):
            assert rel(o[0, row].float(), ro[0, row]) < 2 * tol, (row, rel(o[0, row].float(), ro[0, row]))
        assert (lse.double() - rl).abs().max() < (1e-3 if P == L.PREC_F32 else 0.25), (lse.double() - rl).abs().max()


def test_attn_fwd_fast_equals_exact(dev):
    """inside the window the optimistic kernel and the exact (running-max) kernel agree to bf16 rounding."""
    from hsenet_amd import functional as Fn, _lib as L
    lib = L.load()
    qkv = rnd((2, 700, 3 * 128), 33, dev, 1.2).bfloat16()
    o_fast, lse_fast = Fn.attn_fwd(qkv, 2, L.PREC_BF16, q_prescaled=True)
    lib.hs_debug_set(0, 1)
    try:
        o_exact, lse_exact = Fn.attn_fwd(qkv, 2, L.PREC_BF16, q_prescaled=True)
    finally:
        lib.hs_debug_set(0, 0)
    assert rel(o_fast.float(), o_exact.float()) < 6e-3
    assert (lse_fast - lse_exact).abs().max() < 1e-4        # same q' in both: only the summation order differs


@pytest.mark.parametrize("N", [9, 64, 65, 200, 513])
def test_attn_bwd(dev, N):
    from hsenet_amd import functional as Fn, _lib as L
    B, H = 2, 2
    qkv = rnd((B, N, 3 * H * 64), 40 + N, dev, 1.2).bfloat16()
    d_o = rnd((B, N, H * 64), 41 + N, dev).bfloat16()
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_BF16)
    dqkv = Fn.attn_bwd(qkv, o, d_o, lse, H)
    x = qkv.double().requires_grad_()
    ro, _ = sdpa_ref(x, H)
    (g,) = torch.autograd.grad((ro * d_o.double()).sum(), x)
    D = H * 64
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        e = rel(dqkv[..., sl].float(), g[..., sl])
        # P, dS and the outputs are rounded to bf16 (2^-9 each), delta uses the bf16-rounded O
        assert e < 1.5e-2, (name, e)


@pytest.mark.parametrize("N", [1, 9, 64, 65, 200, 513, 2049])
def test_attn_bwd_f32(dev, N):
    """the exact-f32 backward (parity mode: v_mfma_f32_32x32x2_f32, no bf16 rounding) vs fp64 autograd: 1e-5."""
    from hsenet_amd import functional as Fn, _lib as L
    B, H = 2, 3
    qkv = rnd((B, N, 3 * H * 64), 70 + N, dev, 1.2)
    d_o = rnd((B, N, H * 64), 71 + N, dev)
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_F32)
    dqkv = Fn.attn_bwd(qkv, o, d_o, lse, H)
    assert dqkv.dtype == torch.float32
    x = qkv.double().requires_grad_()
    ro, _ = sdpa_ref(x, H)
    (g,) = torch.autograd.grad((ro * d_o.double()).sum(), x)
    D = H * 64
    gn = float(g.norm())
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        # (at N = 1 the softmax is constant: dq = dk = 0 exactly and ours is fp32 cancellation noise, ~1e-7 of the gradient's
        # norm; the error is therefore taken relative to at least 5 % of the whole gradient)
        e = float((dqkv[..., sl].double() - g[..., sl]).norm()) / max(float(g[..., sl].norm()), 5e-2 * gn)
        assert e < 1e-5, (name, N, e)
    with pytest.raises(L.HsError):
        Fn.attn_bwd(qkv, o.bfloat16(), d_o, lse, H)              # mixed dtypes are refused


@pytest.mark.parametrize("N", [2049, 4097])
def test_attn_bwd_full_length_vs_fp64(dev, N):
    """the sequence lengths the encoders actually run (native 2049, 256^3 4097) against an fp64 statement of SABlock's core:
    B=2 x H=12 like a real layer; the fp64 reference is evaluated head by head on the GPU (N x N fp64 = 134 MB per head)."""
    from hsenet_amd import functional as Fn, _lib as L
    B, H = 2, 12
    D = H * 64
    qkv = rnd((B, N, 3 * D), 640 + N, dev, 1.1).bfloat16()
    d_o = rnd((B, N, D), 641 + N, dev).bfloat16()
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_BF16)
    dqkv = Fn.attn_bwd(qkv, o, d_o, lse, H).float()
    num, den = torch.zeros(3, dtype=torch.float64, device=dev), torch.zeros(3, dtype=torch.float64, device=dev)
    worst_head = 0.0
    for b in range(B):
        for h in range(H):
            sl = [slice(i * D + h * 64, i * D + (h + 1) * 64) for i in range(3)]
            q, k, v = (qkv[b, :, s].double().requires_grad_() for s in sl)
            p = (q @ k.t() * 0.125).softmax(-1)
            ro = p @ v
            assert rel(o[b, :, h * 64:(h + 1) * 64].float(), ro.detach()) < 6e-3
            gq, gk, gv = torch.autograd.grad((ro * d_o[b, :, h * 64:(h + 1) * 64].double()).sum(), (q, k, v))
            for i, g in enumerate((gq, gk, gv)):
                diff = (dqkv[b, :, sl[i]].double() - g).pow(2).sum()
                num[i] += diff
                den[i] += g.pow(2).sum()
                worst_head = max(worst_head, float((diff / g.pow(2).sum()).sqrt()))
            del q, k, v, p, ro, gq, gk, gv
    errs = (num / den).sqrt().tolist()
    print(f"attn bwd N={N}: dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}; worst single head/tensor {worst_head:.2e}")
    assert max(errs) < 1.5e-2 and worst_head < 2e-2


@pytest.mark.parametrize("rows,d", [(5, 768), (1030, 768), (64, 384), (7, 1024), (4097, 768), (8200, 384)])   # last two: the few-rows
def test_layernorm(dev, rows, d):                                                                               # backward (8 waves, row pairs)
    from hsenet_amd import functional as Fn, _lib as L
    x, r = rnd((rows, d), 50, dev, 2.0) + 0.3, rnd((rows, d), 51, dev)
    g, b = 1 + 0.1 * rnd((d,), 52, dev), rnd((d,), 53, dev)
    y, yb, mean, rstd = Fn.layernorm_fwd(x, g, b, resid=r, want_bf16=True)
    xr = (x.double() + r.double()).requires_grad_()
    gr, br = g.double().requires_grad_(), b.double().requires_grad_()
    ref = F.layer_norm(xr, (d,), gr, br, 1e-5)
    assert rel(y, ref) < 1e-6
    assert rel(yb.float(), ref) < 4e-3
    dy = rnd((rows, d), 54, dev)
    gx, gg, gb = torch.autograd.grad((ref * dy.double()).sum(), (xr, gr, br))
    out = Fn.LayerNormFn.apply(x.requires_grad_(), r.requires_grad_(), g.requires_grad_(), b.requires_grad_(), 1e-5)
    out.backward(dy)
    assert rel(x.grad, gx) < 1e-5 and rel(r.grad, gx) < 1e-5
    assert rel(g.grad, gg) < 1e-5 and rel(b.grad, gb) < 1e-5


@pytest.mark.parametrize("rows", [777, 4097, 12290])          # 4 waves x 1 row / the few-rows variant (odd row count, ragged pairs)
def test_layernorm_bwd_bf16_dy_and_colsum(dev, rows):
    from hsenet_amd import functional as Fn, _lib as L
    d = 768
    x, g, b = rnd((rows, d), 55, dev, 2.0), 1 + 0.1 * rnd((d,), 56, dev), rnd((d,), 57, dev)
    dyb, dx_in = rnd((rows, d), 58, dev).bfloat16(), rnd((rows, d), 59, dev)
    _, _, mean, rstd = Fn.layernorm_fwd(x, g, b)
    dx, dxb = torch.empty_like(x), torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
    dg, db, dsum = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    L.call("hs_layernorm_bwd", dyb.data_ptr(), 1, x.data_ptr(), None, g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx_in.data_ptr(),
           dx.data_ptr(), dxb.data_ptr(), dg.data_ptr(), db.data_ptr(), dsum.data_ptr(), rows, d, torch.cuda.current_stream().cuda_stream)
    xr, gr, br = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
    ref = F.layer_norm(xr, (d,), gr, br, 1e-5)
    gx, gg, gb = torch.autograd.grad((ref * dyb.double()).sum(), (xr, gr, br))
    want = gx + dx_in.double()
    assert rel(dx, want) < 1e-5 and rel(dxb.float(), want) < 4e-3
    assert rel(dg, gg) < 1e-5 and rel(db, gb) < 1e-5 and rel(dsum, want.sum(0)) < 1e-5


@pytest.mark.parametrize("rows", [16388, 32700])        # 4 and ~8 volumes' worth of rows: the few-rows (8-wave) variant is selected
def test_layernorm_bwd_variants_agree_bitwise(dev, rows):
    """the two LayerNorm-backward variants (hand-written one-row path, 4 waves; row pairs through the shared lambdas, 8 waves) must
    not drift apart: the whole call (selects the 8-wave variant at these row counts) against the same rows fed in chunks of 4000
    (< 4096 rows: always the 4-wave variant).  A row's dx depends on that row only -> bit-identical; the column sums differ only
    by the order of their float atomics."""
    from hsenet_amd import functional as Fn, _lib as L
    d, s = 768, torch.cuda.current_stream().cuda_stream
    x, g, b = rnd((rows, d), 65, dev, 2.0), 1 + 0.1 * rnd((d,), 66, dev), rnd((d,), 67, dev)
    dyb, dx_in = rnd((rows, d), 68, dev).bfloat16(), rnd((rows, d), 69, dev)
    _, _, mean, rstd = Fn.layernorm_fwd(x, g, b)

    def run(r0, r1, dx, dxb, sums):
        L.call("hs_layernorm_bwd", dyb[r0:r1].data_ptr(), 1, x[r0:r1].data_ptr(), None, g.data_ptr(), mean[r0:r1].data_ptr(),
               rstd[r0:r1].data_ptr(), dx_in[r0:r1].data_ptr(), dx[r0:r1].data_ptr(), dxb[r0:r1].data_ptr(), sums[0].data_ptr(),
               sums[1].data_ptr(), sums[2].data_ptr(), r1 - r0, d, s)
    outs = []
    for chunk in (rows, 4000):
        dx, dxb = torch.empty_like(x), torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        sums = [torch.zeros(d, device=dev) for _ in range(3)]
        for r0 in range(0, rows, chunk):
            run(r0, min(rows, r0 + chunk), dx, dxb, sums)
        outs.append((dx, dxb, sums))
    (dx_w, dxb_w, s_w), (dx_n, dxb_n, s_n) = outs
    assert torch.equal(dx_w, dx_n) and torch.equal(dxb_w, dxb_n)
    for a, c in zip(s_w, s_n):
        assert rel(a, c) < 1e-5


@pytest.mark.parametrize("geom", [((2, 1, 8, 32, 32), (4, 16, 16)), ((1, 1, 32, 64, 48), (16, 16, 16)), ((1, 2, 4, 6, 10), (2, 3, 5))])
@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_patchify(dev, geom, prec):
    from einops import rearrange
    from hsenet_amd import _lib as L
    shape, ps = geom
    P = L.PREC_BF16 if prec == "bf16" else L.PREC_F32
    vol = rnd(shape, 60, dev)
    B, C, D, H, W = shape
    Np, Pd = (D // ps[0]) * (H // ps[1]) * (W // ps[2]), C * ps[0] * ps[1] * ps[2]
    out = torch.empty(B * Np, Pd, dtype=torch.bfloat16 if prec == "bf16" else torch.float32, device=dev)
    L.call("hs_patchify", vol.data_ptr(), out.data_ptr(), B, C, D, H, W, *ps, P, torch.cuda.current_stream().cuda_stream)
    ref = rearrange(vol, "b c (h p1) (w p2) (d p3) -> (b h w d) (p1 p2 p3 c)", p1=ps[0], p2=ps[1], p3=ps[2])
    ref = ref.bfloat16() if prec == "bf16" else ref
    assert torch.equal(out, ref)          # pure data movement (+ RNE cast): bit-exact


def test_assemble_and_score(dev):
    from hsenet_amd import functional as Fn
    B, Np, d = 2, 24, 768
    pe, pos, cls = rnd((B, Np, d), 70, dev).requires_grad_(), rnd((1, Np, d), 71, dev).requires_grad_(), rnd((1, 1, d), 72, dev).requires_grad_()
    x = rnd((B, Np, d), 73, dev).requires_grad_()
    w, b = (0.05 * rnd((1, d), 74, dev)).requires_grad_(), rnd((1,), 75, dev).requires_grad_()
    score = Fn.RowdotSigmoidFn.apply(x, w, b)
    tok, pe_pos = Fn.AssembleFn.apply(pe, pos, cls, score)
    dy = rnd((B, Np + 1, d), 76, dev)
    tok.backward(dy)
    leaves = [t.detach().double().requires_grad_() for t in (pe, pos, cls, x, w, b)]
    pe2, pos2, cls2, x2, w2, b2 = leaves
    s2 = torch.sigmoid(x2 @ w2.t() + b2).squeeze(-1)
    t2 = torch.cat([cls2.expand(B, -1, -1), (pe2 + pos2) * s2.unsqueeze(-1)], 1)
    assert rel(tok, t2) < 1e-6 and rel(pe_pos, pe2 + pos2) < 1e-6 and rel(score, s2) < 1e-6
    grads = torch.autograd.grad((t2 * dy.double()).sum(), leaves)
    for mine, ref, name in zip((pe, pos, cls, x, w, b), grads, "pe pos cls x w b".split()):
        assert rel(mine.grad, ref) < 1e-5, name


def test_cross_attn(dev):
    from hsenet_amd import functional as Fn
    B, nq, nk, d = 2, 40, 32, 768
    q, k, v = (rnd(s, 80 + i, dev).requires_grad_() for i, s in enumerate(((B, nq, d), (B, nk, d), (B, nk, d))))
    out, p = Fn.CrossAttnFn.apply(q, k, v)
    dy = rnd((B, nq, d), 84, dev)
    out.backward(dy)
    q2, k2, v2 = (t.detach().double().requires_grad_() for t in (q, k, v))
    pr = (q2 @ k2.transpose(1, 2) / math.sqrt(d)).softmax(-1)
    ref = pr @ v2
    assert rel(out, ref) < 1e-5 and rel(p, pr) < 1e-5
    for mine, r, name in zip((q, k, v), torch.autograd.grad((ref * dy.double()).sum(), (q2, k2, v2)), "qkv"):
        assert rel(mine.grad, r) < 1e-5, name


@pytest.mark.parametrize("grid,kernel", [((8, 16, 16), (1, 4, 4)), ((16, 16, 16), (2, 4, 4)), ((4, 4, 4), (2, 2, 2)), ((3, 5, 6), (1, 2, 3))])
def test_pool_and_window_attn(dev, grid, kernel):
    from hsenet_amd import functional as Fn
    B, d = 2, 768
    T = grid[0] * grid[1] * grid[2]
    hr = rnd((B, *grid, d), 90, dev).requires_grad_()
    lr = Fn.Pool3dFn.apply(hr, kernel)
    hr2 = hr.detach().double().requires_grad_()
    lr2 = F.avg_pool3d(hr2.permute(0, 4, 1, 2, 3), kernel_size=kernel).permute(0, 2, 3, 4, 1).reshape(B, -1, d)
    assert rel(lr, lr2) < 1e-6
    dl = rnd(tuple(lr.shape), 91, dev)
    lr.backward(dl)
    (g,) = torch.autograd.grad((lr2 * dl.double()).sum(), hr2)
    assert rel(hr.grad, g) < 1e-6
    if any(g_ % k_ for g_, k_ in zip(grid, kernel)):
        return
    cells = lr.shape[1]
    q = rnd((B, cells, d), 92, dev).requires_grad_()
    kv = rnd((B, T, 2 * d), 93, dev).requires_grad_()
    out = Fn.WindowAttnFn.apply(q, kv, grid, kernel)
    dy = rnd((B, cells, d), 94, dev)
    out.backward(dy)
    q2, kv2 = q.detach().double().requires_grad_(), kv.detach().double().requires_grad_()
    kd, kh, kw = kernel
    cd, ch, cw = grid[0] // kd, grid[1] // kh, grid[2] // kw

    def windows(t):                                     # [B, T, d] -> [B, cells, nwin, d]
        t = t.view(B, cd, kd, ch, kh, cw, kw, d).permute(0, 1, 3, 5, 2, 4, 6, 7)
        return t.reshape(B, cells, kd * kh * kw, d)
    kk, vv = windows(kv2[..., :d]), windows(kv2[..., d:])
    p = (q2.unsqueeze(2) @ kk.transpose(-1, -2) / math.sqrt(d)).softmax(-1)
    ref = (p @ vv).squeeze(2)
    assert rel(out, ref) < 1e-5
    gq, gkv = torch.autograd.grad((ref * dy.double()).sum(), (q2, kv2))
    assert rel(q.grad, gq) < 1e-5 and rel(kv.grad, gkv) < 1e-5


def test_linear_fn_and_misc(dev):
    from hsenet_amd import functional as Fn, _lib as L
    x = rnd((3, 50, 768), 100, dev).requires_grad_()
    w, b = (0.03 * rnd((1536, 768), 101, dev)).requires_grad_(), rnd((1536,), 102, dev).requires_grad_()
    dy = rnd((3, 50, 1536), 103, dev)
    for P, tol in ((L.PREC_F32, 1e-5), (L.PREC_BF16, 6e-3)):
        for t in (x, w, b):
            t.grad = None
        y = Fn.LinearFn.apply(x, w, b, P)
        y.backward(dy)
        x2, w2, b2 = (t.detach().double().requires_grad_() for t in (x, w, b))
        ref = F.linear(x2, w2, b2)
        assert rel(y, ref) < tol
        for mine, r, name in zip((x, w, b), torch.autograd.grad((ref * dy.double()).sum(), (x2, w2, b2)), "xwb"):
            assert rel(mine.grad, r) < tol, (name, rel(mine.grad, r))
    # gelu / l2norm
    z = rnd((1000,), 104, dev).requires_grad_()
    g = Fn.GeluFn.apply(z)
    g.backward(torch.ones_like(g))
    z2 = z.detach().double().requires_grad_()
    r = F.gelu(z2)
    assert rel(g, r) < 1e-6 and rel(z.grad, torch.autograd.grad(r.sum(), z2)[0]) < 1e-5
    u = rnd((9, 768), 105, dev).requires_grad_()
    n = Fn.L2NormFn.apply(u, 1e-12)
    dn = rnd((9, 768), 106, dev)
    n.backward(dn)
    u2 = u.detach().double().requires_grad_()
    rn = F.normalize(u2, dim=-1)
    assert rel(n, rn) < 1e-6 and rel(u.grad, torch.autograd.grad((rn * dn.double()).sum(), u2)[0]) < 1e-5
    # colsum + casts
    m = rnd((777, 768), 107, dev)
    out = torch.zeros(768, device=dev)
    Fn.colsum(m.bfloat16(), out)
    assert rel(out, m.bfloat16().double().sum(0)) < 1e-5
    assert torch.equal(Fn.cast_bf16(m), m.bfloat16()) and torch.equal(Fn.cast_f32(m.bfloat16()), m.bfloat16().float())


def test_adamw(dev):
    from hsenet_amd import functional as Fn
    n = 10007
    p, g = rnd((n,), 110, dev), rnd((n,), 111, dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pr = p.clone().requires_grad_()
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    for step in (1, 2, 3):
        pr.grad = g.clone() * step
        opt.step()
        Fn.adamw_step(p, g * step, m, v, lr=1e-2, weight_decay=0.05, step=step)
    assert rel(p, pr.detach()) < 1e-6


@pytest.mark.parametrize("n", [4, 8, 80])
def test_clip_head_vs_oracle(dev, n):
    """N1: cls projection + normalize + symmetric InfoNCE (CLIP_stage1.py:97-102,141-155) incl. all gradients."""
    from hsenet_amd import clip, _lib as L
    from oracle import ref_cpu as R
    d = 768
    tokens = rnd((n, 5, d), 300, dev)
    proj = torch.nn.Linear(d, d).to(dev)
    txt = F.normalize(rnd((n, d), 301, dev), dim=-1).requires_grad_()
    scale = torch.tensor(float(np.log(1 / 0.07)), device=dev, requires_grad=True)
    tokens.requires_grad_()
    img = clip.encode_image_cls(tokens, proj, L.PREC_F32)
    loss, lpi, lpt = clip.image_text_contrastive_learning(img, txt, scale)
    loss.backward()
    t2 = tokens.detach().double().cpu().requires_grad_()
    p2 = torch.nn.Linear(d, d).double()
    p2.load_state_dict({k: v.double().cpu() for k, v in proj.state_dict().items()})
    txt2 = txt.detach().double().cpu().requires_grad_()
    s2 = scale.detach().double().cpu().requires_grad_()
    img2 = F.normalize(p2(t2), dim=-1)[:, 0]          # the reference projects every token, then keeps [:, 0]
    ref = R.clip_contrastive_loss(img2, txt2, s2)
    ref.backward()
    assert rel(img, img2) < 1e-5 and abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert rel(lpi, s2.detach() * img2.detach() @ txt2.detach().t()) < 1e-5 and torch.equal(lpt, lpi.t())
    assert rel(tokens.grad, t2.grad) < 1e-4 and rel(txt.grad, txt2.grad) < 1e-4
    assert rel(proj.weight.grad, p2.weight.grad) < 1e-4 and rel(scale.grad, s2.grad) < 1e-4


# ---- size-independent properties at the full 256^3 geometry (N = 4097, 12 heads) --------------------------------------
# the fp64 reference is too slow at this size; these hold for ANY exact softmax attention and are checked on the kernels
@pytest.mark.gpu
def test_attn_full_size_properties(dev):
    from hsenet_amd import functional as Fn, _lib as L
    B, H, N, D = 2, 12, 4097, 768
    g = torch.Generator(device="cpu").manual_seed(77)
    qkv = (torch.randn(B, N, 3 * D, generator=g) * 1.2).to(dev).bfloat16()
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_BF16)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    # (1) the output is a convex combination of the values: per (batch, head, channel) it lies inside the value range
    v = qkv[..., 2 * D:].float()
    assert (o.float() <= v.amax(1, keepdim=True) + 1e-2).all() and (o.float() >= v.amin(1, keepdim=True) - 1e-2).all()
    # (2) permuting the keys/values (same permutation, every batch and head) changes nothing: softmax has no key order
    perm = torch.randperm(N, generator=g).to(dev)
    qkv_p = qkv.clone()
    qkv_p[..., D:] = qkv[:, perm][..., D:]
    o_p, lse_p = Fn.attn_fwd(qkv_p, H, L.PREC_BF16)
    assert rel(o_p.float(), o.float()) < 4e-3                    # bf16 P and O rounding, different summation order
    assert (lse_p - lse).abs().max() < 1e-4
    # (3) linear in V: o(2 v) = 2 o(v) exactly (a power of two commutes with every rounding)
    qkv_2 = qkv.clone()
    qkv_2[..., 2 * D:] = qkv[..., 2 * D:] * 2
    o_2, _ = Fn.attn_fwd(qkv_2, H, L.PREC_BF16)
    assert torch.equal(o_2.float(), 2 * o.float())
    # (4) shifting every score of a row by a constant (k -> k + c q_hat is not constant; use q -> q, k -> k, add to lse only):
    #     lse rows are >= the largest score and <= largest score + ln N
    q, k = qkv[..., :D].float().view(B, N, H, 64), qkv[..., D:2 * D].float().view(B, N, H, 64)
    rows = torch.tensor([0, 1, 2048, 4096], device=dev)
    s = torch.einsum("bqhd,bkhd->bhqk", q[:, rows], k) * 0.125
    smax = s.amax(-1)
    got = lse[:, :, rows]
    assert (got >= smax - 2e-2).all() and (got <= smax + torch.log(torch.tensor(float(N))) + 2e-2).all()


@pytest.mark.gpu
def test_attn_bwd_full_size_properties(dev):
    from hsenet_amd import functional as Fn, _lib as L
    B, H, N, D = 1, 12, 4097, 768
    g = torch.Generator(device="cpu").manual_seed(78)
    qkv = (torch.randn(B, N, 3 * D, generator=g) * 1.1).to(dev).bfloat16()
    d_o = torch.randn(B, N, D, generator=g).to(dev).bfloat16()
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_BF16)
    dqkv = Fn.attn_bwd(qkv, o, d_o, lse, H)
    assert torch.isfinite(dqkv.float()).all()
    # (1) linear in dO: a power-of-two scale commutes with every rounding in the backward
    dqkv2 = Fn.attn_bwd(qkv, o, (d_o.float() * 2).bfloat16(), lse, H)
    assert torch.equal(dqkv2.float(), 2 * dqkv.float())
    # (2) dV = P^T dO, so the column sums of dV over the keys equal the column sums of dO over the queries (rows of P sum to 1)
    dv = dqkv[..., 2 * D:].float().sum(1)
    assert rel(dv, d_o.float().sum(1)) < 1e-2
    # (3) softmax backward: sum_k dS[q,k] = 0, hence sum_k dK[k,:] = sum_q (sum_k dS[q,k]) ... = 0 only through q; check the
    #     identity that needs no reference: q . dq = k . dk summed over everything (both equal sum_{q,k} dS S / scale)
    qf, kf = qkv[..., :D].float(), qkv[..., D:2 * D].float()
    lhs = (qf * dqkv[..., :D].float()).sum()
    rhs = (kf * dqkv[..., D:2 * D].float()).sum()
    assert abs(float(lhs - rhs)) < 2e-2 * max(abs(float(lhs)), abs(float(rhs)), 1.0)


# ---- train-mode dropout (the reference's hard-coded Dropout(0.1) sites): statistics, seed replay, forward/backward consistency ----
@pytest.mark.gpu
def test_dropout_kernel_statistics_and_replay(dev):
    from hsenet_amd import functional as Fn
    x = torch.ones(1 << 20, device=dev)
    y1 = Fn.DropoutFn.apply(x, 0.1, 1234, 7)
    y2 = Fn.DropoutFn.apply(x, 0.1, 1234, 7)
    y3 = Fn.DropoutFn.apply(x, 0.1, 1235, 7)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)                  # a pure function of (seed, offset, index)
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3                                            # 1e6 Bernoulli(0.9) draws: sigma = 3e-4
    assert torch.allclose(y1[y1 != 0], torch.tensor(1 / 0.9, device=dev))
    assert abs(y1.mean().item() - 1.0) < 3e-3                                # unbiased
    # neighbouring elements are independent (a counter RNG must not repeat within a 4-element Philox block)
    a, b = (y1[0::2] != 0).float(), (y1[1::2] != 0).float()
    assert abs(((a - a.mean()) * (b - b.mean())).mean().item()) < 2e-3


@pytest.mark.gpu
def test_train_mode_dropout_sites(dev):
    """regular_attention / resolution_attention_v3 in train(): seed-replayable, unbiased around the eval output, and the
    backward uses the forward's masks (checked against finite differences of the SAME masked function)."""
    import hsenet_amd as H
    torch.manual_seed(0)
    ra = H.regular_attention().to(dev)
    q, kv = torch.randn(2, 64, 768, device=dev), torch.randn(2, 32, 768, device=dev)
    ra.eval()
    y_eval, p_eval = ra(q, kv, kv)
    ra.train()
    torch.manual_seed(5); y_a, p_a = ra(q, kv, kv)
    torch.manual_seed(5); y_b, p_b = ra(q, kv, kv)
    torch.manual_seed(6); y_c, _ = ra(q, kv, kv)
    assert torch.equal(y_a, y_b) and torch.equal(p_a, p_b) and not torch.equal(y_a, y_c)
    assert not torch.equal(y_a, y_eval)
    assert abs((p_a == 0).float().mean().item() - 0.1) < 0.02                # ~4096 probabilities, 10 % dropped
    assert torch.allclose(p_a[p_a != 0], (p_eval / 0.9)[p_a != 0], rtol=1e-5, atol=1e-7)
    # packer side
    pa = H.resolution_attention_v3().to(dev)
    hr = torch.randn(1, 4, 8, 8, 768, device=dev)
    lr = hr.view(1, 4, 2, 4, 2, 4, 768).mean((3, 5))
    pa.train()
    torch.manual_seed(9); z1 = pa(lr, hr, (1, 4, 4))
    torch.manual_seed(9); z2 = pa(lr, hr, (1, 4, 4))
    pa.eval()
    z0 = pa(lr, hr, (1, 4, 4))
    assert torch.equal(z1, z2) and not torch.equal(z1, z0)


@pytest.mark.gpu
def test_dropout_backward_uses_forward_mask(dev):
    from hsenet_amd import functional as Fn
    g = torch.Generator(device="cpu").manual_seed(3)
    q = torch.randn(1, 8, 64, generator=g).to(dev).requires_grad_()
    k = torch.randn(1, 16, 64, generator=g).to(dev).requires_grad_()
    v = torch.randn(1, 16, 64, generator=g).to(dev).requires_grad_()
    w = torch.randn(1, 8, 64, generator=g).to(dev)
    out, pd = Fn.CrossAttnFn.apply(q, k, v, 0.25, 77, 3)
    (out * w).sum().backward()
    # the same masked function in torch: mask recovered from the returned dropped map
    qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
    p = torch.softmax(qd @ kd.transpose(1, 2) / 8.0, -1)
    mask = (pd != 0).double() / 0.75
    ref = (p * mask) @ vd
    (ref * w.double()).sum().backward()
    assert rel(out, ref.detach()) < 1e-5
    for a, b in ((q.grad, qd.grad), (k.grad, kd.grad), (v.grad, vd.grad)):
        assert rel(a, b) < 1e-4, rel(a, b)
    # window attention: mask is internal -> compare the analytic gradient with a central difference of the masked forward
    qw = torch.randn(1, 2, 64, generator=g).to(dev)
    kvw = torch.randn(1, 32, 128, generator=g).to(dev)
    ww = torch.randn(1, 2, 64, generator=g).to(dev)
    qg = qw.clone().requires_grad_()
    o = Fn.WindowAttnFn.apply(qg, kvw, (2, 4, 4), (1, 4, 4), 0.25, 99, 1)
    (o * ww).sum().backward()
    eps = 1e-2
    d = torch.zeros_like(qw); d[0, 1, 5] = eps
    f = lambda t: (Fn.WindowAttnFn.apply(t, kvw, (2, 4, 4), (1, 4, 4), 0.25, 99, 1) * ww).sum().item()
    fd = (f(qw + d) - f(qw - d)) / (2 * eps)
    assert abs(fd - qg.grad[0, 1, 5].item()) < 2e-3 * max(1.0, abs(fd)), (fd, qg.grad[0, 1, 5].item())


# ---- the 256x256-tile kernels (grid >= 230 blocks): transposed-accumulator register epilogue (NT), staged epilogue (NN), ragged last
# row tile, streaming stores on (>= 64 MiB of output) and off ----
@pytest.mark.gpu
def test_gemm_big_tile_paths(dev):
    from hsenet_amd import functional as Fn, _lib as L
    M, N, K = 16400, 1024, 192                       # 65 x 4 tiles, the last row tile holds 16 rows
    a = rnd((M, K), 21, dev).bfloat16()
    w = rnd((N, K), 22, dev).bfloat16()
    bias, resid = rnd((N,), 23, dev), rnd((M, N), 24, dev)
    acc = (a.double() @ w.double().t())
    # NT, no flags, the first 256 columns carry an extra factor (the qkv GEMM's q third); 33 MB of output: ordinary stores
    o16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    Fn.gemm(a, w, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16, scale_cols=256, scale_cols_alpha=0.25)
    want = acc.clone(); want[:, :256] *= 0.25
    # one rounding of an fp32 sum: equal to the rounded fp64 result except where the summation order tips a tie (a few per million)
    assert (o16 != want.float().bfloat16()).float().mean().item() < 1e-3 and rel(o16.float(), want) < 3e-3
    # NT bias + residual -> fp32, 67 MB: streaming stores
    o32 = torch.full((M, N), float("nan"), device=dev)
    Fn.gemm(a, w, L.GEMM_NT, prec=L.PREC_BF16, out_f32=o32, bias=bias, resid=resid, flags=L.EPI_BIAS | L.EPI_RESID)
    assert rel(o32, acc + bias.double() + resid.double()) < 1e-6
    # NT bias + GELU, keeping gelu'(z): two bf16 outputs, 67 MB together
    g16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    Fn.gemm(a, w, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16, bias=bias, aux_out=g16, flags=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_DGELU)
    z = (acc + bias.double()).float().bfloat16().float()                 # autocast: linear1 emits bf16
    zq = z.double().requires_grad_()
    (gq,) = torch.autograd.grad(F.gelu(zq).sum(), zq)
    assert rel(o16.float(), F.gelu(z)) < 3e-3 and rel(g16.float(), gq) < 3e-3
    assert not torch.isnan(o16.float()).any() and not torch.isnan(g16.float()).any()
    # NN (dgrad): dy [M, N] @ w [N, K'] with K' = 1024 wide output needs its own operands
    dy = rnd((M, 256), 25, dev).bfloat16()
    w2 = rnd((256, N), 26, dev).bfloat16()                                # stored [K=256][N]
    acc2 = dy.double() @ w2.double()
    Fn.gemm(dy, w2, L.GEMM_NN, prec=L.PREC_BF16, out_f32=o32)
    assert rel(o32, acc2) < 1e-6
    Fn.gemm(dy, w2, L.GEMM_NN, prec=L.PREC_BF16, out_bf16=o16, aux=g16, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU)
    assert rel(o16.float(), acc2 * g16.double()) < 3e-3
    # the same dgrad against the TRANSPOSED weight copy = an NT GEMM on the 16x16x32 kernel, x gelu'(z), fused column sums
    # (what hs_vit_blocks_bwd launches for fc2's dgrad); must agree with the NN kernel's result and the fp64 statement
    w2t = w2.t().contiguous()                                             # [N][K=256]
    o16t = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    cs, cs_nn = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    Fn.gemm(dy, w2t, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16t, aux=g16, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU, colsum_out=cs)
    Fn.gemm(dy, w2, L.GEMM_NN, prec=L.PREC_BF16, out_bf16=o16, aux=g16, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU, colsum_out=cs_nn)
    want2 = acc2 * g16.double()
    assert rel(o16t.float(), want2) < 3e-3 and (o16t != o16).float().mean().item() < 1e-3
    assert rel(cs, want2.sum(0)) < 1e-5 and rel(cs_nn, want2.sum(0)) < 1e-5
    Fn.gemm(dy, w2t, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16t, aux=g16, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU)        # no column sums
    assert rel(o16t.float(), want2) < 3e-3
