"""Tensor-level wrappers and autograd Functions over the C ABI (hsenet_amd/_lib.py).

torch is plumbing here (device memory, streams, autograd graph); all arithmetic runs in libhsenet_hip.so.
Every entry point requires CUDA(ROCm) tensors and raises otherwise — there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib as L

_PREC = {"bf16": L.PREC_BF16, "f32": L.PREC_F32, "f16": L.PREC_F16}
_DT16 = {L.PREC_BF16: torch.bfloat16, L.PREC_F16: torch.float16}
_default_precision = "bf16"


def set_precision(p: str) -> None:
    """'bf16' (default; the reference's autocast policy: bf16 MFMA inputs, fp32 accumulate/residual) or
    'f32' (exact-f32 MFMA parity mode: forward and backward, no bf16 rounding anywhere), or 'f16' (the same kernels with IEEE fp16
    operand packs at the bf16 kernels' rate, forward AND backward since round 5 -- the reference's eval / generation scripts run under
    torch.cuda.amp.autocast(), default dtype fp16, Bench/eval/eval_HSENet_CT_Rate_MRG.py:383,423.  fp16 has 5 exponent bits: a
    backward pass wants its loss multiplied by a power of two so that the activation gradients sit in fp16's range, exactly as
    torch.cuda.amp.GradScaler does for an fp16 autocast step -- the engines do it themselves (loss_scale), a module-API caller
    scales the loss and unscales the gradients)."""
    global _default_precision
    if p not in _PREC:
        raise ValueError(f"precision must be one of {list(_PREC)}")
    _default_precision = p


def get_precision() -> str:
    return _default_precision


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.HsError("hsenet_amd kernels need tensors on an MI355X ('cuda') device; there is no CPU fallback")
        if not t.is_contiguous():
            raise L.HsError("hsenet_amd kernels need contiguous tensors")


def _need_as(t: torch.Tensor, what: str, shape, dtype=torch.float32) -> None:
    """entry-point check of a tensor whose raw pointer goes to a kernel that reads prod(shape) elements of `dtype`: a wrong
    dtype or shape would be an out-of-bounds device read (or silently wrong numbers), not an exception."""
    if not isinstance(t, torch.Tensor):
        raise L.HsError(f"{what} must be a tensor, got {type(t).__name__}")
    _need(t)
    if t.dtype != dtype or tuple(t.shape) != tuple(shape):
        raise L.HsError(f"{what} must be {dtype} with shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.float32 else t.float()


# ----------------------------------------------------------------------------------------------------
# raw ops
# ----------------------------------------------------------------------------------------------------
def gemm(A: torch.Tensor, Bm: torch.Tensor, layout: int, *, prec: int, out_f32: Optional[torch.Tensor] = None,
         out_bf16: Optional[torch.Tensor] = None, bias=None, resid=None, aux=None, aux_out=None, colsum_out=None,
         alpha: float = 1.0, flags: int = 0, split_k: int = 1, scale_cols: int = 0, scale_cols_alpha: float = 1.0) -> None:
    """C[M,N] = sum_k A(m,k) B(n,k); 2-D or batched 3-D operands (see include/hsenet_hip.h)."""
    _need(A, Bm, out_f32, out_bf16, bias, resid, aux, aux_out, colsum_out)
    want = _DT16.get(prec, torch.float32)
    if A.dtype != want or Bm.dtype != want:
        raise L.HsError(f"gemm operands must be {want}")
    batch = A.shape[0] if A.dim() == 3 else 1
    a2, b2 = A.shape[-2:], Bm.shape[-2:]
    if layout == L.GEMM_NT:
        M, K, N, K2 = a2[0], a2[1], b2[0], b2[1]
    elif layout == L.GEMM_NN:
        M, K, K2, N = a2[0], a2[1], b2[0], b2[1]
    else:
        K, M, K2, N = a2[0], a2[1], b2[0], b2[1]
    if K != K2:
        raise L.HsError(f"gemm inner dims differ: {K} vs {K2}")
    out = out_f32 if out_f32 is not None else out_bf16
    if out is None or tuple(out.shape[-2:]) != (M, N):
        raise L.HsError("gemm output shape mismatch")
    g = L.GemmArgs()
    g.A, g.B, g.M, g.N, g.K = A.data_ptr(), Bm.data_ptr(), M, N, K
    g.lda, g.ldb, g.batch = a2[1], b2[1], batch
    g.strideA = a2[0] * a2[1] if A.dim() == 3 else 0
    g.strideB = b2[0] * b2[1] if Bm.dim() == 3 else 0
    g.strideC = M * N
    g.out_f32, g.out_bf16, g.ldc = _p(out_f32), _p(out_bf16), N
    g.bias, g.resid, g.aux, g.aux_out, g.colsum_out = _p(bias), _p(resid), _p(aux), _p(aux_out), _p(colsum_out)
    g.alpha, g.flags, g.layout, g.split_k = alpha, flags, layout, split_k
    g.scale_cols, g.scale_cols_alpha = scale_cols, scale_cols_alpha
    L.call("hs_gemm", C.byref(g), prec, _s())


def wgrad_group(dys, xs, outs, split_k: int = 1) -> None:
    """outs[i] += dys[i]^T @ xs[i] for up to 4 (dY [rows][n_out], X [rows][n_in]) bf16 (or all-fp16) pairs in ONE launch
    (hs_gemm_group_prec); outs: fp32 [n_out][n_in], accumulated with float atomics (the caller zeroes them)."""
    n = len(dys)
    dt = dys[0].dtype if n else torch.bfloat16
    if not (1 <= n <= 4) or len(xs) != n or len(outs) != n:
        raise L.HsError("wgrad_group takes 1..4 (dY, X, out) triples")
    arr = (L.GemmArgs * n)()
    for g, dy, x, o in zip(arr, dys, xs, outs):
        _need(dy, x, o)
        if dt not in (torch.bfloat16, torch.float16) or dy.dtype != dt or x.dtype != dt or o.dtype != torch.float32:
            raise L.HsError("wgrad_group: bf16 (or all fp16) operands, fp32 outputs")
        if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != x.shape[0] or tuple(o.shape) != (dy.shape[1], x.shape[1]):
            raise L.HsError("wgrad_group: shape mismatch")
        g.A, g.B, g.M, g.N, g.K = dy.data_ptr(), x.data_ptr(), dy.shape[1], x.shape[1], dy.shape[0]
        g.lda, g.ldb, g.ldc, g.batch = dy.shape[1], x.shape[1], x.shape[1], 1
        g.out_f32, g.alpha, g.flags, g.layout, g.split_k = o.data_ptr(), 1.0, L.EPI_ATOMIC, L.GEMM_TN, split_k
    L.call("hs_gemm_group_prec", arr, n, L.PREC_F16 if dt == torch.float16 else L.PREC_BF16, _s())


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    _need(x)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.call("hs_cast_bf16", x.data_ptr(), y.data_ptr(), x.numel(), _s())
    return y


def cast16(x: torch.Tensor, prec: int) -> torch.Tensor:
    """fp32 -> the 16-bit operand format of `prec` (bf16 / fp16), round to nearest even."""
    if prec == L.PREC_BF16:
        return cast_bf16(x)
    if prec != L.PREC_F16:
        raise L.HsError("cast16: the precision has no 16-bit operand format")
    _need(x)
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    L.call("hs_cast_f16", x.data_ptr(), y.data_ptr(), x.numel(), _s())
    return y


def cast_f32(x: torch.Tensor) -> torch.Tensor:
    _need(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.call("hs_cast_f32", x.data_ptr(), y.data_ptr(), x.numel(), _s())
    return y


def colsum(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[n] += sum_m x[m, n] (x 2-D bf16, fp16 or fp32)."""
    _need(x, out)
    fmt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[x.dtype]
    L.call("hs_colsum", x.data_ptr(), fmt, out.data_ptr(), x.shape[0], x.shape[1], x.shape[1], _s())


def _wgrad_split(Mo: int, No: int, K: int, tile=(256, 128), kt: int = 64) -> int:
    """split-K for a wgrad: minimise (rounds over 256 CUs) x (K-tiles per block + fixed per-block cost)."""
    tiles = ((Mo + tile[0] - 1) // tile[0]) * ((No + tile[1] - 1) // tile[1])
    nkt = (K + kt - 1) // kt
    best, best_cost = 1, None
    for s in range(1, min(32, nkt) + 1):
        cost = ((tiles * s + 255) // 256) * ((nkt + s - 1) // s + 16)
        if best_cost is None or cost < best_cost:
            best, best_cost = s, cost
    return best


def attn_fwd(qkv: torch.Tensor, H: int, prec: int, q_prescaled: bool = False):
    """qkv [B,N,3*H*64] -> (o [B,N,H*64], lse [B,H,N]).  q_prescaled: the q third already carries dh^-0.5*log2(e)."""
    _need(qkv)
    if qkv.dtype != _DT16.get(prec, torch.float32):
        raise L.HsError(f"attn_fwd: qkv must be {_DT16.get(prec, torch.float32)} at this precision, got {qkv.dtype}")
    B, N, D3 = qkv.shape
    o = torch.empty(B, N, D3 // 3, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    L.call("hs_attn_fwd", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, 64, prec,
           L.ATTN_Q_PRESCALED if q_prescaled else 0, _s())
    return o, lse


def attn_bwd(qkv, o, d_o, lse, H: int, q_prescaled: bool = False):
    """dqkv from (qkv, o, dO, lse): bf16 tensors, fp16 ones, or fp32 ones (exact-f32 parity mode, raw q)."""
    _need(qkv, o, d_o, lse)
    B, N, _ = qkv.shape
    if not (qkv.dtype == o.dtype == d_o.dtype) or qkv.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise L.HsError("attn_bwd: qkv, o, dO must all be bf16, all fp16 or all fp32")
    flags = ((L.ATTN_Q_PRESCALED if q_prescaled else 0) | (L.ATTN_F32 if qkv.dtype == torch.float32 else 0)
             | (L.ATTN_F16 if qkv.dtype == torch.float16 else 0))
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(2, B, H, N, dtype=torch.float32, device=qkv.device)   # row constants -delta | -lse*log2e
    L.call("hs_attn_bwd", qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(),
           B, N, H, 64, flags, _s())
    return dqkv


def layernorm_fwd(x, gamma, beta, *, resid=None, want_f32=True, want_bf16=False, eps=1e-5):
    _need(x, gamma, beta, resid)
    rows, d = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x) if want_f32 else None
    yb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    L.call("hs_layernorm_fwd", x.data_ptr(), _p(resid), gamma.data_ptr(), beta.data_ptr(), _p(y), _p(yb), mean.data_ptr(),
           rstd.data_ptr(), rows, d, eps, _s())
    return y, yb, mean, rstd


def adamw_step(p, g, m, v, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale=1.0, step=1) -> None:
    _need(p, g, m, v)
    L.call("hs_adamw", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1], eps,
           weight_decay, grad_scale, step, _s())


# ----------------------------------------------------------------------------------------------------
# autograd Functions (each mirrors one reference operator; file:line in the docstrings)
# ----------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """nn.Linear (y = x W^T + b) on the MFMA GEMM; x fp32 [..., K] -> fp32 [..., N]."""

    @staticmethod
    def forward(ctx, x, weight, bias, prec):
        x = _f32(x).contiguous()
        _need(x, weight, bias)
        K, N = x.shape[-1], weight.shape[0]
        x2 = x.view(-1, K)
        y = torch.empty(x2.shape[0], N, dtype=torch.float32, device=x.device)
        if prec != L.PREC_F32:
            xa, wa = cast16(x2, prec), cast16(weight, prec)
        else:
            xa, wa = x2, weight.contiguous()
        gemm(xa, wa, L.GEMM_NT, prec=prec, out_f32=y, bias=bias, flags=L.EPI_BIAS if bias is not None else 0)
        ctx.save_for_backward(xa, wa)
        ctx.prec, ctx.has_bias, ctx.xshape = prec, bias is not None, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xa, wa = ctx.saved_tensors
        prec = ctx.prec
        dy2 = _f32(dy).contiguous().view(-1, wa.shape[0])
        dya = cast16(dy2, prec) if prec != L.PREC_F32 else dy2
        M, N, K = xa.shape[0], wa.shape[0], wa.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
            gemm(dya, wa, L.GEMM_NN, prec=prec, out_f32=dx)
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros(N, K, dtype=torch.float32, device=dy.device)
            tile, kt = ((256, 128), 64) if prec != L.PREC_F32 else ((64, 64), 32)
            gemm(dya, xa, L.GEMM_TN, prec=prec, out_f32=dw, flags=L.EPI_ATOMIC, split_k=_wgrad_split(N, K, M, tile, kt))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(N, dtype=torch.float32, device=dy.device)
            colsum(dya, db)
        return dx, dw, db, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of (x + resid) (vit.py:62: norm(query_list + x))."""

    @staticmethod
    def forward(ctx, x, resid, gamma, beta, eps):
        x = _f32(x).contiguous()
        resid = None if resid is None else _f32(resid).contiguous()
        y, _, mean, rstd = layernorm_fwd(x, gamma, beta, resid=resid, eps=eps)
        ctx.save_for_backward(x, resid, gamma, mean, rstd)
        ctx.has_resid = resid is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, resid, gamma, mean, rstd = ctx.saved_tensors
        dy = _f32(dy).contiguous()
        _need(dy)
        rows, d = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        L.call("hs_layernorm_bwd", dy.data_ptr(), 0, x.data_ptr(), _p(resid), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
               None, dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), None, rows, d, _s())
        return dx, (dx if ctx.has_resid else None), dg, db, None


class GeluFn(torch.autograd.Function):
    """nn.GELU() exact erf (spatial_pooling_projector.py:129)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x).contiguous()
        _need(x)
        y = torch.empty_like(x)
        L.call("hs_gelu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), _s())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32(dy).contiguous()
        dx = torch.empty_like(x)
        L.call("hs_gelu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _s())
        return dx


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) (CLIP_stage1.py:101)."""

    @staticmethod
    def forward(ctx, x, eps):
        x = _f32(x).contiguous()
        _need(x)
        y = torch.empty_like(x)
        L.call("hs_l2norm_fwd", x.data_ptr(), y.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], eps, _s())
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32(dy).contiguous()
        dx = torch.empty_like(x)
        L.call("hs_l2norm_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], ctx.eps, _s())
        return dx, None


class PatchEmbedFn(torch.autograd.Function):
    """MONAI PatchEmbeddingBlock's rearrange + Linear (vit.py:428-437,455): vol [B,C,D,H,W] fp32 ->
    [B, Np, hidden] fp32 (bias included, position embedding NOT — AssembleFn adds it)."""

    @staticmethod
    def forward(ctx, vol, weight, bias, patch_size, prec):
        vol = _f32(vol).contiguous()
        _need(vol, weight, bias)
        B, Cc, D, H, W = vol.shape
        p1, p2, p3 = patch_size
        Np, Pd, d = (D // p1) * (H // p2) * (W // p3), Cc * p1 * p2 * p3, weight.shape[0]
        dt = _DT16.get(prec, torch.float32)
        patches = torch.empty(B * Np, Pd, dtype=dt, device=vol.device)
        L.call("hs_patchify", vol.data_ptr(), patches.data_ptr(), B, Cc, D, H, W, p1, p2, p3, prec, _s())
        wa = cast16(weight, prec) if prec != L.PREC_F32 else weight.contiguous()
        y = torch.empty(B * Np, d, dtype=torch.float32, device=vol.device)
        gemm(patches, wa, L.GEMM_NT, prec=prec, out_f32=y, bias=bias, flags=L.EPI_BIAS)
        ctx.save_for_backward(patches)
        ctx.prec, ctx.wshape = prec, weight.shape
        return y.view(B, Np, d)

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        prec = ctx.prec
        N, K = ctx.wshape
        dy2 = _f32(dy).contiguous().view(-1, N)
        dya = cast16(dy2, prec) if prec != L.PREC_F32 else dy2
        dw = torch.zeros(N, K, dtype=torch.float32, device=dy.device)
        tile, kt = ((256, 128), 64) if prec != L.PREC_F32 else ((64, 64), 32)
        gemm(dya, patches, L.GEMM_TN, prec=prec, out_f32=dw, flags=L.EPI_ATOMIC,
             split_k=_wgrad_split(N, K, patches.shape[0], tile, kt))
        db = torch.zeros(N, dtype=torch.float32, device=dy.device)
        colsum(dya, db)
        return None, dw, db, None, None


class AssembleFn(torch.autograd.Function):
    """tokens = cat(cls, (pe + pos) * score) (vit.py:345-349,459-461).  Returns (tokens, pe+pos)."""

    @staticmethod
    def forward(ctx, pe, pos, cls, score):
        pe = pe.contiguous()
        _need(pe, pos, cls, score)
        B, Np, d = pe.shape
        tokens = torch.empty(B, Np + 1, d, dtype=torch.float32, device=pe.device)
        pe_pos = torch.empty_like(pe)
        L.call("hs_assemble_tokens", pe.data_ptr(), pos.data_ptr(), cls.data_ptr(), _p(score), tokens.data_ptr(), pe_pos.data_ptr(),
               B, Np, d, _s())
        ctx.save_for_backward(pe_pos, score)
        ctx.shapes = (pos.shape, cls.shape)
        ctx.mark_non_differentiable(pe_pos)
        return tokens, pe_pos

    @staticmethod
    def backward(ctx, dtok, _unused):
        pe_pos, score = ctx.saved_tensors
        dtok = _f32(dtok).contiguous()
        B, Np, d = pe_pos.shape
        dpe = torch.empty_like(pe_pos)
        dpos = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=dtok.device)
        dcls = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dtok.device)
        dscore = torch.empty(B, Np, dtype=torch.float32, device=dtok.device) if score is not None else None
        L.call("hs_assemble_tokens_bwd", dtok.data_ptr(), pe_pos.data_ptr(), _p(score), dpe.data_ptr(), dpos.data_ptr(), dcls.data_ptr(),
               _p(dscore), B, Np, d, _s())
        return dpe, dpos, dcls, dscore


def draw_dropout_seed():
    """(seed, offset) for one dropout site of one forward call, from torch's default CPU generator (so ``torch.manual_seed``
    replays the masks).  The mask itself is Philox4x32-10 inside the kernels: a pure function of (seed, offset, element)."""
    v = torch.randint(0, 2 ** 62, (2,), dtype=torch.int64)
    return int(v[0]), int(v[1]) & 0xFFFFFFFF


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in train mode with a replayable Philox mask (hs_dropout); identity is handled by the caller."""

    @staticmethod
    def forward(ctx, x, p, seed, offset):
        x = _f32(x).contiguous()
        _need(x)
        y = torch.empty_like(x)
        L.call("hs_dropout", x.data_ptr(), y.data_ptr(), x.numel(), float(p), seed, offset, _s())
        ctx.args = (float(p), seed, offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset = ctx.args
        dy = _f32(dy).contiguous()
        dx = torch.empty_like(dy)
        L.call("hs_dropout", dy.data_ptr(), dx.data_ptr(), dy.numel(), p, seed, offset, _s())
        return dx, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """attention(query, key, value) of regular_attention (vit.py:25-33): returns (dropout(P) V, dropout(P)); drop_p = 0 in
    eval mode (then the second output is P itself)."""

    @staticmethod
    def forward(ctx, q, k, v, drop_p=0.0, seed=0, offset=0):
        q, k, v = (_f32(t).contiguous() for t in (q, k, v))
        _need(q, k, v)
        B, nq, d = q.shape
        nk = k.shape[1]
        out = torch.empty_like(q)
        p = torch.empty(B, nq, nk, dtype=torch.float32, device=q.device)
        pd = torch.empty_like(p) if drop_p > 0 else None
        L.call("hs_cross_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), p.data_ptr(), _p(pd), B, nq, nk, d,
               float(drop_p), seed, offset, _s())
        ctx.save_for_backward(q, k, v, p, *([pd] if pd is not None else []))
        ctx.drop = (float(drop_p), seed, offset)
        ret = pd if pd is not None else p
        ctx.mark_non_differentiable(ret)
        return out, ret

    @staticmethod
    def backward(ctx, dout, _dp):
        q, k, v, p, *rest = ctx.saved_tensors
        pd = rest[0] if rest else None
        drop_p, seed, offset = ctx.drop
        dout = _f32(dout).contiguous()
        B, nq, d = q.shape
        nk = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty_like(p)
        L.call("hs_cross_attn_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), p.data_ptr(), _p(pd), dout.data_ptr(), dq.data_ptr(),
               dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, nq, nk, d, drop_p, seed, offset, _s())
        return dq, dk, dv, None, None, None


class RowdotSigmoidFn(torch.autograd.Function):
    """patch_score_proj (Linear(d,1)) + Sigmoid (vit.py:338-339): x [..., d] -> [...]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _f32(x).contiguous()
        w = w.contiguous()
        _need(x, w, b)
        rows, d = x.numel() // x.shape[-1], x.shape[-1]
        score = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        L.call("hs_rowdot_sigmoid_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), score.data_ptr(), rows, d, _s())
        ctx.save_for_backward(x, w, score)
        return score

    @staticmethod
    def backward(ctx, dscore):
        x, w, score = ctx.saved_tensors
        dscore = _f32(dscore).contiguous()
        rows, d = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        db = torch.zeros(1, dtype=torch.float32, device=x.device)
        L.call("hs_rowdot_sigmoid_bwd", x.data_ptr(), w.data_ptr(), score.data_ptr(), dscore.data_ptr(), dx.data_ptr(), dw.data_ptr(),
               db.data_ptr(), rows, d, _s())
        return dx, dw, db


class Pool3dFn(torch.autograd.Function):
    """F.avg_pool3d over the token grid (spatial_pooling_projector.py:141): hr [B,gd,gh,gw,d] -> [B,cells,d]."""

    @staticmethod
    def forward(ctx, hr, kernel):
        hr = _f32(hr).contiguous()
        _need(hr)
        B, gd, gh, gw, d = hr.shape
        kd, kh, kw = kernel
        cells = (gd // kd) * (gh // kh) * (gw // kw)
        lr = torch.empty(B, cells, d, dtype=torch.float32, device=hr.device)
        L.call("hs_pool3d_fwd", hr.data_ptr(), lr.data_ptr(), B, gd, gh, gw, kd, kh, kw, d, _s())
        ctx.geom = (B, gd, gh, gw, d, kd, kh, kw)
        return lr

    @staticmethod
    def backward(ctx, dlr):
        B, gd, gh, gw, d, kd, kh, kw = ctx.geom
        dlr = _f32(dlr).contiguous()
        dhr = torch.empty(B, gd, gh, gw, d, dtype=torch.float32, device=dlr.device)
        L.call("hs_pool3d_bwd", dlr.data_ptr(), dhr.data_ptr(), B, gd, gh, gw, kd, kh, kw, d, 0, _s())
        return dhr, None


class WindowAttnFn(torch.autograd.Function):
    """1 query per LR cell vs its kd*kh*kw HR children (spatial_pooling_projector.py:68-76); drop_p > 0 = train-mode dropout on
    the probabilities (Philox mask, replayed in backward).
    q [B,cells,d]; kv [B,tokens,2d] (k|v)."""

    @staticmethod
    def forward(ctx, q, kv, grid, kernel, drop_p=0.0, seed=0, offset=0):
        q, kv = _f32(q).contiguous(), _f32(kv).contiguous()
        _need(q, kv)
        B, cells, d = q.shape
        out = torch.empty_like(q)
        L.call("hs_window_attn_fwd", q.data_ptr(), kv.data_ptr(), out.data_ptr(), B, *grid, *kernel, d, float(drop_p), seed, offset, _s())
        ctx.save_for_backward(q, kv)
        ctx.geom = (tuple(grid), tuple(kernel))
        ctx.drop = (float(drop_p), seed, offset)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv = ctx.saved_tensors
        grid, kernel = ctx.geom
        dout = _f32(dout).contiguous()
        B, cells, d = q.shape
        dq = torch.empty_like(q)
        dkv = torch.zeros_like(kv)
        L.call("hs_window_attn_bwd", q.data_ptr(), kv.data_ptr(), dout.data_ptr(), dq.data_ptr(), dkv.data_ptr(), B, *grid, *kernel, d,
               *ctx.drop, _s())
        return dq, dkv, None, None, None, None, None


def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def blocks_forward_inference(x0, H, mlp, prec, params, ln_eps=0.0, qkv_bias=None):
    """Forward-only run of the fused block stack with the options the 2-D trunk needs (LayerNorm eps, biased qkv
    projection: hs_vit_desc.ln_eps / .qkv_bias).  params: L*11 block tensors in the driver's order + norm.weight, norm.bias."""
    x0 = _f32(x0).contiguous()
    params = [p.contiguous() for p in params]
    _need(x0, *params)
    B, N, d = x0.shape
    Ln = (len(params) - 2) // 11
    qb = None
    if qkv_bias is not None:
        qkv_bias = [b.contiguous() for b in qkv_bias]
        _need(*qkv_bias)
        if len(qkv_bias) != Ln:
            raise L.HsError("one qkv bias per layer expected")
        qb = _ptr_array(qkv_bias)
    desc = L.VitDesc(B, N, d, H, mlp, Ln, prec, 0, float(ln_eps), C.cast(qb, C.c_void_p) if qb is not None else None)
    nbytes = L.load().hs_vit_workspace_bytes(C.byref(desc))
    if nbytes == 0:
        raise L.HsError(f"unsupported encoder geometry B={B} N={N} d={d} H={H} mlp={mlp} L={Ln}")
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x0.device)
    off = (-ws.data_ptr()) % 256
    y = torch.empty_like(x0)
    hidden = torch.empty(Ln, B, N, d, dtype=torch.float32, device=x0.device)
    L.call("hs_vit_blocks_fwd", C.byref(desc), x0.data_ptr(), _ptr_array(params[:-2]), _ptr_array(params[-2:]), y.data_ptr(),
           hidden.data_ptr(), ws.data_ptr() + off, nbytes, _s())
    return y


def _aligned_bytes(nbytes: int, device) -> "tuple[torch.Tensor, int]":
    """a byte workspace whose used range starts on a 256-byte boundary (the drivers' requirement): (tensor, offset)."""
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return ws, (-ws.data_ptr()) % 256


def _flat_grads(tensors: Sequence[Optional[torch.Tensor]]):
    """ONE zeroed fp32 buffer holding a gradient per tensor (each on a 64-element boundary) and the views of it in the tensors' shapes
    (None stays None): one memset instead of one torch.zeros_like launch per parameter -- 134 per block stack.  autograd's AccumulateGrad
    adopts such a view as the parameter's .grad as it stands (dense, matching strides) when .grad is None, and adds it in place otherwise."""
    offs, off = [], 0
    for t in tensors:
        offs.append(off)
        if t is not None:
            off += (t.numel() + 63) // 64 * 64
    dev = next(t.device for t in tensors if t is not None)
    flat = torch.zeros(max(off, 64), dtype=torch.float32, device=dev)
    return [None if t is None else flat[o:o + t.numel()].view(t.shape) for t, o in zip(tensors, offs)]


# ---- when do 16-bit weight copies go stale?  Three witnesses, all cheap to read at every forward:
#  (1) torch's version counters: in-place writes through the parameter or a detach()ed alias (load_state_dict, state_dict()[k].copy_(),
#      p.mul_(), foreach optimizers);
#  (2) optimizer steps: torch.optim's FUSED kernels (fused=True) update parameters WITHOUT touching the version counter, so a global
#      optimizer post-step hook stamps every parameter the optimizer owns with a generation number;
#  (3) invalidate_weight_copies(): for writes neither of them sees (`p.data.copy_()`-style edits through an alias with its own version
#      counter, raw-pointer writes); HS_WEIGHT_CACHE=0 turns the caching off altogether (every forward re-casts, as before round 6).
_step_gen = 0
_forced_gen = 0
_param_gen: dict = {}


def _optimizer_stepped(opt, args, kwargs) -> None:
    global _step_gen
    _step_gen += 1
    for g in opt.param_groups:
        for p in g["params"]:
            _param_gen[id(p)] = _step_gen


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook  # noqa: E402

_reg_post_hook(_optimizer_stepped)


def invalidate_weight_copies() -> None:
    """call after changing encoder weights in a way torch does not record (``p.data.mul_()``, a raw-pointer write): every cached 16-bit
    weight copy of the module API and of the engines is re-made at its next forward."""
    global _forced_gen
    _forced_gen += 1


def weights_stamp(mats: Sequence[torch.Tensor]):
    """changes whenever one of `mats` may have changed (see above)."""
    return (sum(w._version for w in mats), sum(_param_gen.get(id(w), 0) for w in mats), _forced_gen, mats[0].data_ptr(), mats[-1].data_ptr())


class W16Cache:
    """Per-module 16-bit copies of the block-stack weights for the MODULE API (hs_vit_desc.w16 / .w16t): made when a forward finds the
    fp32 parameters changed (weights_stamp: version counters, optimizer steps, explicit invalidation), not on every call, and shared by
    every forward in between (eval loops, the frozen towers of the MLLM finetune, gradient accumulation).  One set per operand format.
    A backward pass can never meet copies newer than its forward's: the fp32 parameters are saved tensors of the same node, and torch
    refuses to run a backward whose saved tensors were modified in place (an optimizer step between forward and backward is not a
    thing)."""

    def __init__(self):
        self._sets = {}

    def get(self, desc, mats: Sequence[torch.Tensor], lp, prec: int, want_t: bool):
        """-> (w16, w16t or None) current for `mats` (the 4 weight matrices per layer; `lp` is the driver's pointer table of the L*11
        block parameters)."""
        key = weights_stamp(mats)
        if os.environ.get("HS_WEIGHT_CACHE") == "0":
            key = object()                                     # never equal: re-cast on every call
        ent = self._sets.get(prec)
        if ent is None or ent[0].device != mats[0].device:
            n = L.load().hs_vit_weights_elems(C.byref(desc))
            if n == 0:
                raise L.HsError("unsupported encoder geometry")
            ent = [torch.empty(n, dtype=_DT16[prec], device=mats[0].device), None, None, None]       # w16, w16t, key of w16, key of w16t
            self._sets[prec] = ent
        want_t = want_t and desc.d % 64 == 0 and desc.mlp % 64 == 0
        if want_t and ent[1] is None:
            ent[1] = torch.empty_like(ent[0])
        need_s, need_t = ent[2] != key, want_t and ent[3] != key
        if need_s or need_t:
            if need_s:          # (the transposed set is re-made along with the straight one whenever it exists: one call)
                L.call("hs_vit_cast_weights", C.byref(desc), lp, ent[0].data_ptr(), _p(ent[1]), _s())
                ent[2] = key
                ent[3] = key if ent[1] is not None else None
            else:
                L.call("hs_vit_cast_weights", C.byref(desc), None, ent[0].data_ptr(), ent[1].data_ptr(), _s())
                ent[3] = key
        return ent[0], (ent[1] if want_t else None)


class BlocksFn(torch.autograd.Function):
    """The 12 x TransformerBlock hot loop + final norm (vit.py:464-467 / 352-355) as ONE fused enqueue.

    inputs: x0 [B,N,d] fp32, then L*11 block params in MONAI state-dict order, then norm.weight, norm.bias.
    returns (y [B,N,d], hidden [L,B,N,d]).  Gradients flow through y only (the reference's callers never
    differentiate the hidden-state list).  cache: the module's W16Cache (16-bit precisions) or None."""

    @staticmethod
    def forward(ctx, x0, H, mlp, prec, need_grad, cache, n_qb, *params):
        x0 = _f32(x0).contiguous()
        params = tuple(p.contiguous() for p in params)
        _need(x0, *params)
        B, N, d = x0.shape
        qbias = params[len(params) - n_qb:] if n_qb else ()        # (qkv_bias=True encoders: one [3d] bias per layer, behind the norm)
        params = params[:len(params) - n_qb] if n_qb else params
        Ln = (len(params) - 2) // 11
        if n_qb not in (0, Ln):
            raise L.HsError("one qkv bias per layer expected")
        # need_grad is decided by the caller (grad mode is off inside Function.forward, and
        # ctx.needs_input_grad ignores torch.no_grad())
        qb_arr = _ptr_array(qbias) if n_qb else None
        desc = L.VitDesc(B, N, d, H, mlp, Ln, prec, int(need_grad), 0.0, C.cast(qb_arr, C.c_void_p) if n_qb else None)
        lp, npar = _ptr_array(params[:-2]), _ptr_array(params[-2:])
        w16 = w16t = None
        if cache is not None and prec != L.PREC_F32:
            mats = [params[l * 11 + i] for l in range(Ln) for i in (8, 6, 0, 2)]           # qkv, out_proj, linear1, linear2
            w16, w16t = cache.get(desc, mats, lp, prec, bool(need_grad))
            desc.w16, desc.w16t = w16.data_ptr(), _p(w16t)
        nbytes = L.load().hs_vit_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            raise L.HsError(f"unsupported encoder geometry B={B} N={N} d={d} H={H} mlp={mlp} L={Ln}")
        ws, off = _aligned_bytes(nbytes, x0.device)
        y = torch.empty_like(x0)
        hidden = torch.empty(Ln, B, N, d, dtype=torch.float32, device=x0.device)
        L.call("hs_vit_blocks_fwd", C.byref(desc), x0.data_ptr(), lp, npar, y.data_ptr(), hidden.data_ptr(), ws.data_ptr() + off,
               nbytes, _s())
        if need_grad:
            ctx.save_for_backward(x0, hidden, ws, *params, *qbias)
            ctx.desc, ctx.ws_off, ctx.nbytes, ctx.n_qb, ctx.qb_arr = desc, off, nbytes, n_qb, qb_arr
            ctx.w16 = (w16, w16t)            # keeps the copies this graph was built on alive (a later precision switch makes new sets)
        ctx.mark_non_differentiable(hidden)
        return y, hidden

    @staticmethod
    def backward(ctx, dy, _dh):
        x0, hidden, ws, *params = ctx.saved_tensors
        desc, n_qb = ctx.desc, ctx.n_qb
        dy = _f32(dy).contiguous()
        dx0 = torch.empty_like(x0)
        grads = _flat_grads(params)
        nblk = len(params) - n_qb
        lp, npar = _ptr_array(params[:nblk - 2]), _ptr_array(params[nblk - 2:nblk])
        lg, ng = _ptr_array(grads[:nblk - 2]), _ptr_array(grads[nblk - 2:nblk])
        if n_qb:
            qg = _ptr_array(grads[nblk:])
            desc.qkv_bias_grad = C.cast(qg, C.c_void_p)
        L.call("hs_vit_blocks_bwd", C.byref(desc), x0.data_ptr(), lp, npar, dy.data_ptr(), dx0.data_ptr(), lg, ng, hidden.data_ptr(),
               ws.data_ptr() + ctx.ws_off, ctx.nbytes, desc.L, 0, _s())
        desc.qkv_bias_grad = None
        return (dx0, None, None, None, None, None, None, *grads)


def _opt_ptr_array(tensors: Sequence[Optional[torch.Tensor]], n: int):
    arr = (C.c_void_p * n)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _set_dropout(desc, train: bool) -> None:
    """the two hard-coded Dropout(0.1) sites of regular_attention / resolution_attention_v3 (vit.py:46-47, spatial_pooling_projector.py:
    58-59): in train() fresh Philox (seed, offset) pairs -- [0] the probabilities, [1] the pre-residual output, drawn in that order like
    the unfused Functions do; eval(): off."""
    if train:
        desc.drop_p = 0.1
        for i in range(2):
            sd, of = draw_dropout_seed()
            desc.seed[i], desc.offset[i] = sd, of
    else:
        desc.drop_p = 0.0


def patchify16(vol: torch.Tensor, patch_size, prec: int) -> torch.Tensor:
    """the rearranged patch matrix [B*Np, Pd] in the 16-bit operand format (hs_patchify): the A operand of the patch-embed Linear."""
    vol = _f32(vol).contiguous()
    _need(vol)
    B, Cc, D, H, W = vol.shape
    p1, p2, p3 = patch_size
    Np, Pd = (D // p1) * (H // p2) * (W // p3), Cc * p1 * p2 * p3
    patches = torch.empty(B * Np, Pd, dtype=_DT16[prec], device=vol.device)
    L.call("hs_patchify", vol.data_ptr(), patches.data_ptr(), B, Cc, D, H, W, p1, p2, p3, prec, _s())
    return patches


class shared_patches:
    """``with shared_patches(): ...`` -- inside the block every encoder that embeds the SAME volume tensor (the dual tower runs the 3D-ViT
    and the 2E3-ViT on one volume, vit.py:926-930) reads one patch matrix: the volume is patchified once.  Keyed by the identity of the
    volume tensor object + patch size + precision; nothing outlives the block."""
    _active = None

    def __enter__(self):
        self._prev, shared_patches._active = shared_patches._active, {}
        return self

    def __exit__(self, *exc):
        shared_patches._active = self._prev
        return False

    @staticmethod
    def get(vol: torch.Tensor, patch_size, prec: int) -> torch.Tensor:
        tab = shared_patches._active
        if tab is None:
            return patchify16(vol, patch_size, prec)
        key = (id(vol), vol._version, tuple(patch_size), prec)
        hit = tab.get(key)
        if hit is None or hit[0] is not vol:
            hit = (vol, patchify16(vol, patch_size, prec))
            tab[key] = hit
        return hit[1]


class PackerFn(torch.autograd.Function):
    """P2 VisualPacker_3d_phi_v3.forward (spatial_pooling_projector.py:138-146, incl. resolution_attention_v3 :62-83 and proj_mpls) as ONE
    node over the fused packer driver: x [B, T, d] fp32 -> [B, cells, out_dim] fp32.  x may be the reference's `features[:, 1:]` view of an
    encoder output (vit.py:934-936): the driver reads it in place (row offset + batch stride), no copy.  params: the 14 state-dict-order
    tensors [proj_mpls.0.w, .0.b, proj_mpls.2.w, .2.b, Wq.w, .b, Wk.w, .b, Wv.w, .b, output_linear.w, .b, norm.w, .b].  16-bit precisions."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        grid, kernel, out_dim, prec, train, need_grad = meta
        x = _f32(x)
        B, T, d = x.shape
        gd, gh, gw = grid
        kd, kh, kw = kernel
        if T != gd * gh * gw:
            raise L.HsError(f"packer grid {tuple(grid)} does not match {T} input tokens")
        off = 0
        if not x.is_contiguous():
            # rows [off, off + T) of a dense [B, off + T, d] tensor (the `[:, 1:]` patch selection)?  then read it where it lies
            s0, s1, s2 = x.stride()
            if s2 == 1 and s1 == d and s0 > T * d and (s0 - T * d) % d == 0 and x.storage_offset() >= s0 - T * d:
                off = (s0 - T * d) // d
            else:
                x = x.contiguous()
        params = tuple(p.contiguous() for p in params)
        _need(*params)
        if not x.is_cuda:
            raise L.HsError("hsenet_amd kernels need tensors on an MI355X ('cuda') device; there is no CPU fallback")
        cells = (gd // kd) * (gh // kh) * (gw // kw)
        desc = L.PackerDesc(B, gd, gh, gw, kd, kh, kw, d, out_dim, off, (off + T) * d, cells * out_dim, int(need_grad), 0.0)
        desc.prec = prec
        _set_dropout(desc, train)
        nbytes = L.load().hs_packer_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            raise L.HsError("unsupported packer geometry")
        ws, wo = _aligned_bytes(nbytes, x.device)
        out = torch.empty(B, cells, out_dim, dtype=torch.float32, device=x.device)
        L.call("hs_packer_fwd", C.byref(desc), x.data_ptr() - 4 * off * d, _ptr_array(params), out.data_ptr(), ws.data_ptr() + wo, nbytes, _s())
        if need_grad:
            ctx.save_for_backward(ws, *params)
            ctx.desc, ctx.ws_off, ctx.nbytes, ctx.geom = desc, wo, nbytes, (B, T, d, off)
        return out

    @staticmethod
    def backward(ctx, dout):
        ws, *params = ctx.saved_tensors
        B, T, d, off = ctx.geom
        dout = _f32(dout).contiguous()
        _need(dout)
        grads = _flat_grads(params)
        dfeat = torch.empty(B, off + T, d, dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
        L.call("hs_packer_bwd", C.byref(ctx.desc), dout.data_ptr(), _ptr_array(params), _ptr_array(grads), _p(dfeat),
               ws.data_ptr() + ctx.ws_off, ctx.nbytes, _s())
        return (None if dfeat is None else dfeat[:, off:], None, *grads)
