"""N1: the CLIP loss head that consumes the encoder's cls token — MI355X mirror of
M3DCLIP_stage1.encode_image / image_text_contrastive_learning (Preprint/LaMed/src/model/CLIP_stage1.py:97-102,141-155)
and of utils/dist_utils.py:280-306's gather_features.

* ``encode_image_cls``: the reference projects ALL N tokens through ``mm_vision_proj`` and normalises them, then keeps
  ``[:, 0]`` (CLIP_stage1.py:100-101,117); only the cls row is computed here (same values, N-1 rows of GEMM saved).
* ``image_text_contrastive_learning``: logits = logit_scale * img @ txt^T with the RAW ``logit_scale`` parameter as a
  plain multiplier (CLIP_stage1.py:90,148 — it is not exponentiated), symmetric cross-entropy against arange labels.
  Across ranks the features are all-gathered with gradient (one fused collective, hsenet_amd.dp.gather_features).
All arithmetic runs in libhsenet_hip.so (exact-f32 MFMA GEMMs: the matrices are tiny).
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import functional as Fn
from .dp import gather_features


def encode_image_cls(tokens: torch.Tensor, mm_vision_proj: torch.nn.Linear, prec: int = L.PREC_F32) -> torch.Tensor:
    """tokens [B, N, d] (post-norm encoder output) -> L2-normalised projected cls feature [B, d]."""
    cls = tokens[:, 0].contiguous()
    return Fn.L2NormFn.apply(Fn.LinearFn.apply(cls, mm_vision_proj.weight, mm_vision_proj.bias, prec), 1e-12)


class _ClipLossFn(torch.autograd.Function):
    """loss, logits_per_image from (image_features [n,d], text_features [n,d], logit_scale []); labels = arange(n)."""

    @staticmethod
    def forward(ctx, img, txt, scale):
        img, txt = img.float().contiguous(), txt.float().contiguous()
        Fn._need(img, txt, scale)
        n = img.shape[0]
        sc = float(scale)
        logits = torch.empty(n, n, dtype=torch.float32, device=img.device)
        Fn.gemm(img, txt, L.GEMM_NT, prec=L.PREC_F32, out_f32=logits, alpha=sc)
        dlog = torch.empty_like(logits)
        acc = torch.zeros(2, dtype=torch.float32, device=img.device)     # [loss, sum(dlogits * logits)]
        L.call("hs_clip_ce", logits.data_ptr(), dlog.data_ptr(), acc.data_ptr(), n, Fn._s())
        ctx.save_for_backward(img, txt, dlog, acc)
        ctx.sc = sc
        ctx.mark_non_differentiable(logits)
        return acc[0], logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        img, txt, dlog, acc = ctx.saved_tensors
        g = float(dloss)
        dimg = torch.empty_like(img)
        dtxt = torch.empty_like(txt)
        # dL/dimg = s * dlogits @ txt ; dL/dtxt = s * dlogits^T @ img   (exact-f32 MFMA, n x n x d)
        Fn.gemm(dlog, txt, L.GEMM_NN, prec=L.PREC_F32, out_f32=dimg, alpha=ctx.sc * g)
        Fn.gemm(dlog, img, L.GEMM_TN, prec=L.PREC_F32, out_f32=dtxt, alpha=ctx.sc * g)
        dscale = (acc[1] * (g / ctx.sc)).reshape(()) if ctx.needs_input_grad[2] else None     # one scalar multiply (host glue)
        return dimg, dtxt, dscale


def image_text_contrastive_learning(image_features, text_features, logit_scale, gather_loss: bool = True, group=None):
    """(loss, logits_per_image, logits_per_text) as CLIP_stage1.py:141-155 with local_loss=False.
    labels are arange(global batch) as the reference collator builds them (train/train_CLIP_stage1.py:146-151)."""
    if gather_loss:
        image_features, text_features = gather_features(image_features, text_features, group)
    loss, logits = _ClipLossFn.apply(image_features.contiguous(), text_features.contiguous(), logit_scale)
    return loss, logits, logits.t()
