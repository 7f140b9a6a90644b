"""N1: the CLIP loss head that consumes the encoders' cls token — MI355X mirror of the vision half of
M3DCLIP_stage1 / M3DCLIP_stage2 (Preprint/LaMed/src/model/CLIP_stage1.py:56-155, CLIP_stage2.py:58-190) and of
utils/dist_utils.py:280-306's gather_features.  The BERT text towers are out of scope (BASELINE
configs[3] trains against FROZEN text embeddings), so text features are inputs here.

* ``encode_image_cls``: the reference projects ALL N tokens through ``mm_vision_proj`` and normalises them, then keeps
  ``[:, 0]`` (CLIP_stage1.py:100-101,117); only the cls row is computed here (same values, N-1 rows of GEMM saved).
* ``image_text_contrastive_learning``: logits = logit_scale * img @ txt^T with the RAW ``logit_scale`` parameter as a plain
  multiplier (CLIP_stage1.py:90,148 — it is not exponentiated), symmetric cross-entropy against arange labels; both the
  ``local_loss=False`` branch (gathered x gathered) and the ``local_loss=True`` branch (this rank's rows x gathered,
  CLIP_stage2.py:173-177).  Across ranks the features are all-gathered with gradient (hsenet_amd.dp.gather_features).
* ``image_text_relation_regulation`` + ``relation_loss_weight``: the stage-2 relation MSE and its ``0.1 * (1 - step/5000)``
  schedule (CLIP_stage2.py:139-151,187-190).
* ``splice_image_features``: the embedding splice of prepare_inputs_for_multimodal (lamed_arch.py:151-154).
All arithmetic runs in libhsenet_hip.so (exact-f32 MFMA GEMMs: the matrices are tiny); ``logit_scale`` stays on the device.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from .dp import gather_features
from .encoder import ViT_stage1, ViT_stage2


def encode_image_cls(tokens: torch.Tensor, mm_vision_proj: torch.nn.Linear, prec: int = L.PREC_F32) -> torch.Tensor:
    """tokens [B, N, d] (post-norm encoder output) -> L2-normalised projected cls feature [B, d]."""
    cls = tokens[:, 0].contiguous()
    return Fn.L2NormFn.apply(Fn.LinearFn.apply(cls, mm_vision_proj.weight, mm_vision_proj.bias, prec), 1e-12)


class _LogitsFn(torch.autograd.Function):
    """logit_scale * a @ b^T  (a [n,d], b [m,d], logit_scale a 0-d device tensor read ON the device)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = a.float().contiguous(), b.float().contiguous()
        scale = scale.detach().float().reshape(1).contiguous()
        Fn._need(a, b, scale)
        raw = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
        Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_F32, out_f32=raw)
        logits = torch.empty_like(raw)
        L.call("hs_scale_dev", raw.data_ptr(), logits.data_ptr(), scale.data_ptr(), 1.0, raw.numel(), Fn._s())
        ctx.save_for_backward(a, b, raw, scale)
        return logits

    @staticmethod
    def backward(ctx, dlog):
        a, b, raw, scale = ctx.saved_tensors
        dlog = dlog.float().contiguous()
        s = Fn._s()
        draw = torch.empty_like(dlog)
        L.call("hs_scale_dev", dlog.data_ptr(), draw.data_ptr(), scale.data_ptr(), 1.0, dlog.numel(), s)
        da = db = dscale = None
        n, m, d = a.shape[0], b.shape[0], a.shape[1]
        if ctx.needs_input_grad[0]:                  # da = draw @ b      (n x m x d: tiny, any n / m)
            da = torch.empty_like(a)
            L.call("hs_rowmix", draw.data_ptr(), b.data_ptr(), da.data_ptr(), n, m, d, m, d, d, 0, s)
        if ctx.needs_input_grad[1]:                  # db = draw^T @ a
            db = torch.empty_like(b)
            L.call("hs_rowmix", draw.data_ptr(), a.data_ptr(), db.data_ptr(), m, n, d, m, d, d, 1, s)
        if ctx.needs_input_grad[2]:
            dscale = torch.zeros(1, dtype=torch.float32, device=dlog.device)
            L.call("hs_dot", dlog.data_ptr(), raw.data_ptr(), dscale.data_ptr(), 1.0, dlog.numel(), s)
            dscale = dscale.reshape(())
        return da, db, dscale


class _ScaleByFn(torch.autograd.Function):
    """x * g for a 0-d device tensor g (the upstream gradient of a scalar loss), no host read."""

    @staticmethod
    def apply_to(x, g):
        out = torch.empty_like(x)
        g = g.float().reshape(1).contiguous()
        L.call("hs_scale_dev", x.data_ptr(), out.data_ptr(), g.data_ptr(), 1.0, x.numel(), Fn._s())
        return out


class _CeSymFn(torch.autograd.Function):
    """(CE(logits, arange) + CE(logits^T, arange)) / 2 on a square fp32 logits matrix (hs_clip_ce)."""

    @staticmethod
    def forward(ctx, logits):
        logits = logits.float().contiguous()
        Fn._need(logits)
        n = logits.shape[0]
        if logits.shape[1] != n:
            raise L.HsError("symmetric contrastive loss needs a square logits matrix")
        dlog = torch.empty_like(logits)
        acc = torch.zeros(2, dtype=torch.float32, device=logits.device)
        L.call("hs_clip_ce", logits.data_ptr(), dlog.data_ptr(), acc.data_ptr(), n, Fn._s())
        ctx.save_for_backward(dlog)
        return acc[0]

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return _ScaleByFn.apply_to(dlog, g)


class _CeRowsFn(torch.autograd.Function):
    """F.cross_entropy(logits [rows, cols], label0 + arange(rows)) (hs_ce_rows)."""

    @staticmethod
    def forward(ctx, logits, label0):
        logits = logits.float().contiguous()
        Fn._need(logits)
        dlog = torch.empty_like(logits)
        acc = torch.zeros(1, dtype=torch.float32, device=logits.device)
        L.call("hs_ce_rows", logits.data_ptr(), dlog.data_ptr(), acc.data_ptr(), logits.shape[0], logits.shape[1], int(label0), 1.0, 0, Fn._s())
        ctx.save_for_backward(dlog)
        return acc[0]

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return _ScaleByFn.apply_to(dlog, g), None


class _MseFn(torch.autograd.Function):
    """F.mse_loss(ref, x) with gradient to x only (the reference detaches the stage-1 logits, CLIP_stage2.py:139)."""

    @staticmethod
    def forward(ctx, ref, x):
        # mse is invariant to a common transpose: logits_per_text = logits_per_image.T arrives as a transposed view
        tr = (not x.is_contiguous()) and x.dim() == 2 and x.t().is_contiguous()
        if tr:
            ref, x = ref.t(), x.t()
        ref, x = ref.detach().float().contiguous(), x.float().contiguous()
        Fn._need(ref, x)
        d = torch.empty_like(x)
        acc = torch.zeros(1, dtype=torch.float32, device=x.device)
        L.call("hs_relation_mse", ref.data_ptr(), x.data_ptr(), d.data_ptr(), acc.data_ptr(), x.numel(), 1.0, 0, Fn._s())
        ctx.save_for_backward(d)
        ctx.tr = tr
        return acc[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        dx = _ScaleByFn.apply_to(d, g)
        return None, (dx.t() if ctx.tr else dx)


def image_text_contrastive_learning(image_features, text_features, logit_scale, labels=None, gather_loss: bool = True,
                                    local_loss: bool = False, group=None):
    """(loss, logits_per_image, logits_per_text) as CLIP_stage1.py:141-155 / CLIP_stage2.py:170-185.
    ``labels`` must be None: they are arange(global batch) as the reference collator builds them
    (train/train_CLIP_stage1.py:146-151); with ``local_loss`` rank r's rows carry labels r*B_local + arange(B_local)."""
    if labels is not None:
        raise ValueError("labels are implied: arange(global batch), as the reference's DataCollator builds them")
    if gather_loss:
        all_img, all_txt = gather_features(image_features, text_features, group)
        if local_loss:
            on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
            label0 = dist.get_rank(group) * image_features.shape[0] if on else 0
            lpi = _LogitsFn.apply(image_features, all_txt, logit_scale)
            lpt = _LogitsFn.apply(text_features, all_img, logit_scale)
            loss = (_CeRowsFn.apply(lpi, label0) + _CeRowsFn.apply(lpt, label0)) / 2
            return loss, lpi, lpt
        lpi = _LogitsFn.apply(all_img, all_txt, logit_scale)
    else:
        lpi = _LogitsFn.apply(image_features, text_features, logit_scale)
    return _CeSymFn.apply(lpi), lpi, lpi.t()


def image_text_relation_regulation(logits_per_image, logits_per_text, logits_per_image_masked, logits_per_text_masked):
    """CLIP_stage2.py:187-190: (mse(lpi, lpi') + mse(lpt, lpt')) / 2; gradients flow into the primed (stage-2) logits."""
    return (_MseFn.apply(logits_per_image, logits_per_image_masked) + _MseFn.apply(logits_per_text, logits_per_text_masked)) / 2


def relation_loss_weight(global_step, max_weighted_step: int = 5000) -> float:
    """CLIP_stage2.py:141-149: 0.1 * (1 - step/5000) below 5000 steps, 0 after; 0 when the step is unknown (bare except)."""
    try:
        if global_step < max_weighted_step:
            return 0.1 * (1 - global_step / max_weighted_step)
        return 0.0
    except Exception:
        return 0.0


class _SpliceFn(torch.autograd.Function):
    """cat(emb[:, :1], feat, emb[:, n+1:]) along the token axis (lamed_arch.py:153-154) as strided row-block copies."""

    @staticmethod
    def forward(ctx, emb, feat):
        emb, feat = emb.float().contiguous(), feat.float().contiguous()
        Fn._need(emb, feat)
        B, S, d = emb.shape
        n = feat.shape[1]
        if feat.shape[0] != B or feat.shape[2] != d:
            raise L.HsError("image features must be [B, n, hidden] like the token embeddings")
        tail = max(0, S - (n + 1))
        So = 1 + n + tail
        out = torch.empty(B, So, d, dtype=torch.float32, device=emb.device)
        s = Fn._s()
        L.call("hs_rows_scatter", feat.data_ptr(), out.data_ptr(), B, 1, n, So, d, s)          # rows [1, 1+n) <- image tokens
        L.call("hs_rows_copy", emb.data_ptr(), S * d, out.data_ptr(), None, So * d, B, 1, d, s)  # row 0 <- embeds[:, :1]
        if tail:
            L.call("hs_rows_copy", emb.data_ptr() + 4 * (n + 1) * d, S * d, out.data_ptr() + 4 * (n + 1) * d, None, So * d, B, tail, d, s)
        ctx.geom = (B, S, n, d, tail, So)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, n, d, tail, So = ctx.geom
        dout = dout.float().contiguous()
        s = Fn._s()
        dfeat = torch.empty(B, n, d, dtype=torch.float32, device=dout.device)
        L.call("hs_rows_copy", dout.data_ptr() + 4 * d, So * d, dfeat.data_ptr(), None, n * d, B, n, d, s)
        demb = torch.zeros(B, S, d, dtype=torch.float32, device=dout.device)
        L.call("hs_rows_copy", dout.data_ptr(), So * d, demb.data_ptr(), None, S * d, B, 1, d, s)
        if tail:
            L.call("hs_rows_copy", dout.data_ptr() + 4 * (n + 1) * d, So * d, demb.data_ptr() + 4 * (n + 1) * d, None, S * d, B, tail, d, s)
        return demb, dfeat


def splice_image_features(inputs_embeds: torch.Tensor, image_features: torch.Tensor) -> torch.Tensor:
    """prepare_inputs_for_multimodal's splice (lamed_arch.py:151-154): the n projected image tokens replace positions 1..n of
    the token embeddings: cat(embeds[:, :1], image_features, embeds[:, n+1:]).  fp32 in/out, differentiable in both."""
    return _SpliceFn.apply(inputs_embeds, image_features)


# -------------------------------------------------------------------------------------------------------------------------
# the vision halves of the two CLIP wrappers (state-dict keys as the reference: vision_encoder.*, mm_vision_proj.*, logit_scale,
# stage1_pretrained_CLIP.*), text features as inputs
# -------------------------------------------------------------------------------------------------------------------------
def _vit_kwargs(img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads):
    return dict(in_channels=1, img_size=img_size, patch_size=patch_size, hidden_size=hidden_size, mlp_dim=mlp_dim,
                num_layers=num_layers, num_heads=num_heads, pos_embed="perceptron", dropout_rate=0.0, spatial_dims=3, classification=True)


class M3DCLIPVision_stage1(nn.Module):
    """CLIP_stage1.py:56-102 without the BERT tower: ViT_stage1 + mm_vision_proj + logit_scale."""

    def __init__(self, img_size=(32, 256, 256), patch_size=(4, 16, 16), hidden_size=768, mlp_dim=3072, num_layers=12, num_heads=12,
                 gather_loss=True, local_loss=False):
        super().__init__()
        self.vision_encoder = ViT_stage1(**_vit_kwargs(img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads))
        self.mm_vision_proj = nn.Linear(hidden_size, hidden_size)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.gather_loss, self.local_loss = gather_loss, local_loss

    def encode_image(self, image, mask_ratio=None, text_features=None, image_path=None):
        """[B, d]: the cls row of F.normalize(mm_vision_proj(tokens)) — what every caller keeps (CLIP_stage1.py:117)."""
        tokens, _ = self.vision_encoder(image)
        return encode_image_cls(tokens, self.mm_vision_proj)

    def image_text_contrastive_learning(self, image_features, text_features, labels=None, group=None):
        return image_text_contrastive_learning(image_features, text_features, self.logit_scale, labels, self.gather_loss,
                                               self.local_loss, group)

    def forward(self, images, text_features, global_step=None, group=None):
        """CLIP_stage1.py:111-139 with the text tower's output given (text_features: its L2-normalised cls features [B, d])."""
        img = self.encode_image(images)
        loss, lpi, lpt = self.image_text_contrastive_learning(img, text_features, group=group)
        return {"loss": loss, "logits": (lpi + lpt) / 2.0}


class M3DCLIPVision_stage2(nn.Module):
    """CLIP_stage2.py:58-190 without the BERT towers: trainable ViT_stage2 + mm_vision_proj + logit_scale, and the frozen
    stage-1 CLIP.  ``forward`` is the reference's training forward with the text features given."""

    def __init__(self, img_size=(32, 256, 256), patch_size=(4, 16, 16), hidden_size=768, mlp_dim=3072, num_layers=12, num_heads=12,
                 gather_loss=True, local_loss=False):
        super().__init__()
        self.vision_encoder = ViT_stage2(**_vit_kwargs(img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads))
        self.mm_vision_proj = nn.Linear(hidden_size, hidden_size)
        self.stage1_pretrained_CLIP = M3DCLIPVision_stage1(img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads)
        self.stage1_pretrained_CLIP.requires_grad_(False)                      # CLIP_stage2.py:88
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.gather_loss, self.local_loss = gather_loss, local_loss

    def encode_image(self, image, images_2d, mask_ratio=None, text_features=None, image_path=None):
        tokens, _ = self.vision_encoder(image, images_2d)
        return encode_image_cls(tokens, self.mm_vision_proj)

    def image_text_contrastive_learning(self, image_features, text_features, labels=None, group=None):
        return image_text_contrastive_learning(image_features, text_features, self.logit_scale, labels, self.gather_loss,
                                               self.local_loss, group)

    def forward(self, images, text_features_stage1, text_features, images_2d, global_step=None, group=None):
        """CLIP_stage2.py:119-169 with the text towers' outputs given: text_features_stage1 / text_features are the L2-normalised
        cls text features [B, d] of the frozen stage-1 and of the stage-2 text tower."""
        with torch.no_grad():
            img1 = self.stage1_pretrained_CLIP.encode_image(images)
            _, lpi1, lpt1 = self.image_text_contrastive_learning(img1, text_features_stage1, group=group)   # uses self.logit_scale (:128)
        img2 = self.encode_image(images, images_2d)
        loss_cl, lpi2, lpt2 = self.image_text_contrastive_learning(img2, text_features, group=group)
        loss_rel = image_text_relation_regulation(lpi1.detach(), lpt1.detach(), lpi2, lpt2)
        w = relation_loss_weight(global_step)
        return {"loss": loss_cl + w * loss_rel, "logits": (lpi2 + lpt2) / 2.0, "loss_CL_stage2": loss_cl, "loss_relation": loss_rel,
                "relation_loss_weight": w}
