"""Training-step engine for the 3D-ViT encoder (BASELINE configs[1]): forward + backward + DP gradient
all-reduce + AdamW, every kernel from libhsenet_hip.so, parameters / gradients / Adam moments in flat fp32
buffers.  This is the measurement harness' stand-in for the reference's HF Trainer step
(train/train_CLIP_stage1.py:231-244 -> DDP -> optim "adamw_torch" :85); the module API in encoder.py stays the
drop-in surface, and ``FlatEncoder`` adopts such a module's parameters in place (they become views of the flat
buffer, so ``state_dict()`` / ``load_state_dict()`` keep working).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib as L
from . import functional as Fn
from .dp import GradBuckets
from .encoder import ViT_stage1


def _al(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def bucket_plan(names, offsets, total: int, n_layers: int, n_buckets: int):
    """Backward-completion-ordered layer chunks [(l_hi, l_lo)] and the flat-gradient element range each one
    finishes.  Parameters sit in state-dict order (cls, pos, patch-embed, blocks.0 .. blocks.L-1, norm), backward
    completes blocks from L-1 down, so chunk i covers [offset(blocks.l_lo), previous start); the first chunk also
    carries the final norm (it sits behind blocks.L-1), the last one cls/pos/patch-embed (in front of blocks.0)."""
    per = max(1, (n_layers + n_buckets - 1) // n_buckets)
    chunks, hi = [], n_layers
    while hi > 0:
        lo = max(0, hi - per)
        chunks.append((hi, lo))
        hi = lo
    name_off = dict(zip(names, offsets))
    bounds, end = [], total
    for (hi, lo) in chunks:
        start = name_off[f"blocks.{lo}.mlp.linear1.weight"] if lo > 0 else 0
        bounds.append((start, end))
        end = start
    return chunks, bounds


class FlatEncoder:
    """ViT_stage1 with flat parameter / gradient / moment buffers and an explicit fused train step."""

    def __init__(self, model: ViT_stage1, batch: int, *, lr=1e-4, weight_decay=0.1, n_buckets=4, group=None):
        if not isinstance(model, ViT_stage1):
            raise TypeError("FlatEncoder drives ViT_stage1 (BASELINE configs[1])")
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise L.HsError("FlatEncoder needs the module on an MI355X ('cuda') device")
        self.model, self.dev, self.B = model, p0.device, batch
        named = list(model.named_parameters())
        self.names = [n for n, _ in named]
        offs, off = [], 0
        for _, p in named:
            offs.append(off)
            off += _al(p.numel())
        self.offsets, self.total = offs, off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        for (n, p), o in zip(named, offs):
            view = self.flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.lr, self.wd, self.step_no = lr, weight_decay, 0
        pe = model.patch_embedding
        self.Np, self.Pd, self.d = pe.n_patches, pe.patch_dim, model.hidden_size
        self.N = self.Np + 1
        self.Ln = len(model.blocks)
        self.desc = L.VitDesc(batch, self.N, self.d, model.num_heads, model.mlp_dim, self.Ln, L.PREC_BF16, 1, 0.0, None)
        nbytes = L.load().hs_vit_workspace_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise L.HsError("unsupported encoder geometry")
        self.ws_bytes = nbytes
        self.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
        self.ws_ptr = self.ws.data_ptr() + (-self.ws.data_ptr()) % 256
        f32 = dict(dtype=torch.float32, device=self.dev)
        B, Np, N, d = batch, self.Np, self.N, self.d
        self.patches = torch.empty(B * Np, self.Pd, dtype=torch.bfloat16, device=self.dev)
        self.w_pe = torch.empty(d, self.Pd, dtype=torch.bfloat16, device=self.dev)
        self.pe = torch.empty(B * Np, d, **f32)
        self.x0 = torch.empty(B, N, d, **f32)
        self.y = torch.empty(B, N, d, **f32)
        self.hidden = torch.empty(self.Ln, B, N, d, **f32)
        self.dy = torch.empty(B, N, d, **f32)
        self.dx0 = torch.empty(B, N, d, **f32)
        self.dpe = torch.empty(B * Np, d, **f32)
        self.dpe_b = torch.empty(B * Np, d, dtype=torch.bfloat16, device=self.dev)
        blocks = [p for blk in model.blocks for p in blk.ordered_params()]
        self._lp = Fn._ptr_array(blocks)
        self._np = Fn._ptr_array([model.norm.weight, model.norm.bias])
        self._lg = Fn._ptr_array([p.grad for p in blocks])
        self._ng = Fn._ptr_array([model.norm.weight.grad, model.norm.bias.grad])
        self.layer_chunks, bounds = bucket_plan(self.names, offs, self.total, self.Ln, n_buckets)
        self.buckets = GradBuckets(self.grad, bounds, group)

    # -- pieces --------------------------------------------------------------------------------------------
    def forward(self, vol: torch.Tensor) -> torch.Tensor:
        m, s = self.model, Fn._s()
        pe = m.patch_embedding
        lin = pe.patch_embeddings[1]
        B, Cc, D, H, W = vol.shape
        if B != self.B or tuple(vol.shape[2:]) != pe.img_size or not vol.is_cuda or vol.dtype != torch.float32:
            raise L.HsError("FlatEncoder.forward: volume must be fp32 on the GPU with the configured batch/geometry")
        L.call("hs_patchify", vol.data_ptr(), self.patches.data_ptr(), B, Cc, D, H, W, *pe.patch_size, L.PREC_BF16, s)
        L.call("hs_cast_bf16", lin.weight.data_ptr(), self.w_pe.data_ptr(), lin.weight.numel(), s)
        Fn.gemm(self.patches, self.w_pe, L.GEMM_NT, prec=L.PREC_BF16, out_f32=self.pe, bias=lin.bias, flags=L.EPI_BIAS)
        cls = m.cls_token.data_ptr() if hasattr(m, "cls_token") else None
        L.call("hs_assemble_tokens", self.pe.data_ptr(), pe.position_embeddings.data_ptr(), cls, None, self.x0.data_ptr(), None,
               B, self.Np, self.d, s)
        L.call("hs_vit_blocks_fwd", C.byref(self.desc), self.x0.data_ptr(), self._lp, self._np, self.y.data_ptr(),
               self.hidden.data_ptr(), self.ws_ptr, self.ws_bytes, s)
        return self.y

    def backward(self, dy: torch.Tensor) -> None:
        m, s = self.model, Fn._s()
        pe = m.patch_embedding
        lin = pe.patch_embeddings[1]
        self.grad.zero_()
        for i, (hi, lo) in enumerate(self.layer_chunks):
            L.call("hs_vit_blocks_bwd", C.byref(self.desc), self.x0.data_ptr(), self._lp, self._np, dy.data_ptr(), self.dx0.data_ptr(),
                   self._lg, self._ng, self.hidden.data_ptr(), self.ws_ptr, self.ws_bytes, hi, lo, s)
            if lo > 0:
                self.buckets.reduce(i)
        has_cls = hasattr(m, "cls_token")
        L.call("hs_assemble_tokens_bwd", self.dx0.data_ptr(), None, None, self.dpe.data_ptr(), pe.position_embeddings.grad.data_ptr(),
               m.cls_token.grad.data_ptr() if has_cls else None, None, self.B, self.Np, self.d, s)
        L.call("hs_cast_bf16", self.dpe.data_ptr(), self.dpe_b.data_ptr(), self.dpe.numel(), s)
        Fn.gemm(self.dpe_b, self.patches, L.GEMM_TN, prec=L.PREC_BF16, out_f32=lin.weight.grad, flags=L.EPI_ATOMIC,
                split_k=Fn._wgrad_split(self.d, self.Pd, self.B * self.Np))
        Fn.colsum(self.dpe_b, lin.bias.grad)
        self.buckets.reduce(len(self.layer_chunks) - 1)
        self.buckets.wait()

    def optimizer_step(self) -> None:
        self.step_no += 1
        Fn.adamw_step(self.flat, self.grad, self.m, self.v, lr=self.lr, weight_decay=self.wd, grad_scale=self.buckets.grad_scale,
                      step=self.step_no)

    def train_step(self, vol: torch.Tensor) -> None:
        """one step of L = mean(y^2): fwd, dL/dy = 2 y / numel, bwd (+ bucketed all-reduce), AdamW."""
        y = self.forward(vol)
        L.call("hs_scale", y.data_ptr(), self.dy.data_ptr(), 2.0 / y.numel(), y.numel(), Fn._s())
        self.backward(self.dy)
        self.optimizer_step()


def profile_classes() -> List[dict]:
    """read the in-library per-kernel-class timers (hs_prof_*)."""
    lib = L.load()
    out = []
    for i in range(lib.hs_prof_count()):
        cnt, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        L.call("hs_prof_read", i, C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by))
        if cnt.value:
            out.append(dict(name=lib.hs_prof_name(i).decode(), launches=cnt.value, total_ms=ms.value, flops=fl.value, bytes=by.value))
    return out
