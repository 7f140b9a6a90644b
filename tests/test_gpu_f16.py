"""HS_PREC_F16 (GPU): the forward-only fp16 operand mode (the reference's eval / generation scripts run under
torch.cuda.amp.autocast(), default dtype fp16: Bench/eval/eval_HSENet_CT_Rate_MRG.py:383,423).  Kernel level: each fp16 kernel
through the C ABI against a torch fp64 statement on the SAME fp16-rounded operands (so only accumulation order and the output
rounding differ).  The encoder-level fp16 parity against the golden vectors lives in tests/test_gpu_parity.py (prec "f16")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(shape, seed, dev, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy(g.standard_normal(shape, dtype=np.float32) * scale).to(dev)


def test_cast_f16_equals_torch_half_bitwise(dev):
    from hsenet_amd import functional as Fn, _lib as L
    x = rnd((3, 1031), 1, dev, 50.0)
    x[0, :6] = torch.tensor([0.0, -0.0, 65504.0, 65520.0, 1e-8, -7e4], device=dev)          # max, the overflow tie, underflow, -inf
    y = Fn.cast16(x, L.PREC_F16)
    assert y.dtype == torch.float16 and torch.equal(y.view(torch.int16), x.half().view(torch.int16))
    assert torch.equal(Fn.cast16(x, L.PREC_BF16).view(torch.int16), x.bfloat16().view(torch.int16))


# (M, N, K): the 256x256 / 16x16x32 tile (many rows), the 256x128 tile, ragged everything, a K that is no multiple of 64
@pytest.mark.parametrize("M,N,K", [(40970, 768, 768), (4097, 768, 768), (333, 256, 128), (37, 2304, 136), (513, 8, 320)])
def test_gemm_f16_nt_and_forward_epilogues(dev, M, N, K):
    from hsenet_amd import functional as Fn, _lib as L
    a, b = rnd((M, K), 2, dev).half(), rnd((N, K), 3, dev).half()
    bias, resid = rnd((N,), 4, dev), rnd((M, N), 5, dev)
    acc = a.double() @ b.double().t()
    out = torch.full((M, N), float("nan"), device=dev)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F16, out_f32=out)
    assert rel(out, acc) < 1e-5
    o16 = torch.empty(M, N, dtype=torch.float16, device=dev)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F16, out_bf16=o16)                      # (the 16-bit output slot: fp16 in this mode)
    assert rel(o16.float(), acc) < 6e-4 and torch.isfinite(o16).all()            # 2^-11 output rounding
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F16, out_f32=out, bias=bias, resid=resid, flags=L.EPI_BIAS | L.EPI_RESID)
    assert rel(out, acc + bias.double() + resid.double()) < 1e-5
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F16, out_f32=out, bias=bias, flags=L.EPI_BIAS)
    assert rel(out, acc + bias.double()) < 1e-5
    # fc1's epilogue: the pre-activation is rounded to fp16 (what autocast's Linear emits), GELU runs on that
    z = (acc + bias.double()).float().half().double()
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F16, out_bf16=o16, bias=bias, flags=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_DGELU)
    assert rel(o16.float(), torch.nn.functional.gelu(z)) < 8e-4
    # q third pre-scaled (what the fused block stack asks of the qkv projection)
    if N % 12 == 0:
        Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F16, out_f32=out, scale_cols=N // 3, scale_cols_alpha=0.25)
        ref = acc.clone()
        ref[:, :N // 3] *= 0.25
        assert rel(out, ref) < 1e-5


def test_gemm_f16_refuses_everything_but_the_forward(dev):
    from hsenet_amd import functional as Fn, _lib as L
    a, b = rnd((64, 64), 6, dev).half(), rnd((64, 64), 7, dev).half()
    out = torch.empty(64, 64, device=dev)
    z = torch.empty(64, 64, dtype=torch.float16, device=dev)
    for kw in (dict(layout=L.GEMM_NN), dict(layout=L.GEMM_TN, flags=L.EPI_ATOMIC), dict(layout=L.GEMM_NT, flags=L.EPI_GELU_BWD, aux=z),
               dict(layout=L.GEMM_NT, flags=L.EPI_BIAS | L.EPI_GELU, bias=out[0].contiguous(), aux_out=z),
               dict(layout=L.GEMM_NT, colsum_out=out[0].contiguous())):
        layout = kw.pop("layout")
        with pytest.raises(L.HsError):
            Fn.gemm(a, b, layout, prec=L.PREC_F16, out_f32=out, **kw)
    with pytest.raises(L.HsError):                                              # operands of the wrong 16-bit type
        Fn.gemm(a.bfloat16(), b.bfloat16(), L.GEMM_NT, prec=L.PREC_F16, out_f32=out)


@pytest.mark.parametrize("N", [1, 9, 65, 200, 513, 2049])
def test_attn_fwd_f16(dev, N):
    from hsenet_amd import functional as Fn, _lib as L
    B, H = 2, 3
    qkv = rnd((B, N, 3 * H * 64), 8 + N, dev).half()
    q, k, v = (t.view(B, N, H, 64).permute(0, 2, 1, 3).double() for t in qkv.split(H * 64, dim=-1))
    s = q @ k.transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_F16)
    assert o.dtype == torch.float16
    assert rel(o.float(), ref) < 1.5e-3                  # P and the output rounded to fp16 (2^-11)
    assert rel(lse, torch.logsumexp(s, -1)) < 1e-5
    # large logits: exp2(s - max) stays inside fp16 whatever the scale (the un-referenced optimistic kernel is not used in this mode)
    big = (qkv.float() * 6.0).half()
    q, k, v = (t.view(B, N, H, 64).permute(0, 2, 1, 3).double() for t in big.split(H * 64, dim=-1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    o, _ = Fn.attn_fwd(big, H, L.PREC_F16)
    assert torch.isfinite(o).all() and rel(o.float(), ref) < 2e-3


def test_patchify_f16(dev):
    from hsenet_amd import _lib as L, functional as Fn
    for (shape, patch) in (((2, 1, 8, 32, 32), (4, 16, 16)), ((1, 2, 4, 6, 10), (2, 3, 5))):
        vol = rnd(shape, 11, dev)
        B, Cc, D, Hh, W = shape
        p1, p2, p3 = patch
        Np, Pd = (D // p1) * (Hh // p2) * (W // p3), Cc * p1 * p2 * p3
        o16 = torch.empty(B * Np, Pd, dtype=torch.float16, device=dev)
        o32 = torch.empty(B * Np, Pd, dtype=torch.float32, device=dev)
        L.call("hs_patchify", vol.data_ptr(), o16.data_ptr(), B, Cc, D, Hh, W, p1, p2, p3, L.PREC_F16, Fn._s())
        L.call("hs_patchify", vol.data_ptr(), o32.data_ptr(), B, Cc, D, Hh, W, p1, p2, p3, L.PREC_F32, Fn._s())
        assert torch.equal(o16, o32.half())


def test_block_stack_f16_vs_f32_and_backward_refused(dev):
    """hs_vit_blocks_fwd at HS_PREC_F16 against its own exact-f32 mode (2 layers, ViT-B widths); a backward pass through an fp16
    forward raises, and the C entry point refuses save_for_bwd = 1."""
    import ctypes as C
    import hsenet_amd as H
    from hsenet_amd import _lib as L
    kw = dict(in_channels=1, img_size=(8, 64, 64), patch_size=(4, 16, 16), num_layers=2, pos_embed="perceptron", classification=True)
    torch.manual_seed(0)
    m = H.ViT_stage1(**kw).to(dev).eval()
    x = rnd((2, 1, 8, 64, 64), 12, dev)
    out = {}
    for prec in ("f32", "bf16", "f16"):
        H.set_precision(prec)
        try:
            with torch.no_grad():
                out[prec] = m(x)[0]
        finally:
            H.set_precision("bf16")
    e16, eb = rel(out["f16"], out["f32"]), rel(out["bf16"], out["f32"])
    print(f"2-layer stage-1 forward vs exact f32: fp16 {e16:.2e}, bf16 {eb:.2e}")
    assert e16 < 1.5e-3 and e16 < 0.5 * eb              # 2 more mantissa bits on every operand: ~4x closer than bf16
    H.set_precision("f16")
    try:
        with pytest.raises(L.HsError, match="forward-only"):
            y, _ = m(x)
            (y * y).mean().backward()
    finally:
        H.set_precision("bf16")
    desc = L.VitDesc(1, 65, 768, 12, 3072, 1, L.PREC_F16, 1, 0.0, None)
    assert L.load().hs_vit_workspace_bytes(C.byref(desc)) == 0
    desc.save_for_bwd = 0
    assert L.load().hs_vit_workspace_bytes(C.byref(desc)) > 0
