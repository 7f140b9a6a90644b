"""MI355X host mirror of the reference's Spatial Packer projector API
(Preprint/LaMed/src/model/multimodal_projector/{spatial_pooling_projector.py,builder.py}).

Same class names / constructor and forward signatures / ``proj_out_num`` / state-dict keys
(``proj_mpls.{0,2}.*``, ``resolution_attention.{Wq,Wk,Wv,output_linear,norm}.*``).  Arithmetic runs in
libhsenet_hip.so; the modules only hold parameters.  The two hard-coded p=0.1 dropouts of ``resolution_attention_v3``
(spatial_pooling_projector.py:58-59: attention probabilities, pre-residual output) are active in train() — Philox masks
seeded from torch's default CPU generator, regenerated in the backward — and off in eval(); see DESIGN.md "Dropout".
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from ._lib import HsError, PREC_F32
from .encoder import _prec, _unfused


class resolution_attention_v3(nn.Module):
    """spatial_pooling_projector.py:48-83: one query per low-resolution cell attends to the cell's
    high-resolution children (single head, d_k = 768); LN(q + output_linear(attn))."""

    def __init__(self, in_channels=16, out_channels=8, emb_dim=768, output_dim=768, dropout=0.1, aropout=0.0):
        super().__init__()
        self.emb_dim = emb_dim
        self.Wq = nn.Linear(emb_dim, emb_dim)
        self.Wk = nn.Linear(emb_dim, emb_dim)
        self.Wv = nn.Linear(emb_dim, emb_dim)
        self.attn = None
        self.output_linear = nn.Linear(emb_dim, emb_dim)
        self.dropout = nn.Dropout(p=dropout)
        self.dropout_2 = nn.Dropout(p=dropout)
        self.norm = nn.LayerNorm(emb_dim)

    def forward(self, LR_image, HR_image, kernel_size, context=None, mask=None):
        if mask is not None:
            raise ValueError("mask is never passed on the HSENet path (spatial_pooling_projector.py:142)")
        prec = _prec(self)
        B, gd, gh, gw, d = HR_image.shape
        kernel = tuple(int(k) for k in kernel_size)
        lr = LR_image.contiguous().view(B, -1, d)
        q = Fn.LinearFn.apply(lr, self.Wq.weight, self.Wq.bias, prec)
        # K and V projections of every HR token as ONE GEMM against [Wk; Wv]  (59 % of the packer's FLOPs)
        wkv = torch.cat([self.Wk.weight, self.Wv.weight], dim=0)
        bkv = torch.cat([self.Wk.bias, self.Wv.bias], dim=0)
        kv = Fn.LinearFn.apply(HR_image.contiguous().view(B, gd * gh * gw, d), wkv, bkv, prec)
        # hard-coded Dropout(0.1) x2 of the reference, active in train() (spatial_pooling_projector.py:13-14, 58-59, 82)
        train = self.training
        pa, po = (self.dropout.p, self.dropout_2.p) if train else (0.0, 0.0)
        x = Fn.WindowAttnFn.apply(q, kv, (gd, gh, gw), kernel, pa, *(Fn.draw_dropout_seed() if pa > 0 else (0, 0)))
        x = Fn.LinearFn.apply(x, self.output_linear.weight, self.output_linear.bias, prec)
        if po > 0:
            x = Fn.DropoutFn.apply(x, po, *Fn.draw_dropout_seed())
        return Fn.LayerNormFn.apply(q, x, self.norm.weight, self.norm.bias, self.norm.eps)


class VisualPacker_3d_phi_v3(nn.Module):
    """spatial_pooling_projector.py:121-153.  ``grid`` / ``kernel_size`` are keyword-only extensions: the
    reference hard-codes the (8,16,16) token grid (:140) and kernel (1,4,4) (:132); BASELINE's 256^3 configs
    have a (16,16,16) grid, for which the reference-sanctioned kernel is (2,4,4) (:166-167) -> still 128 tokens.
    With the defaults this is the reference module exactly."""

    def __init__(self, image_size, patch_size, in_dim, out_dim, layer_type, layer_num, pooling_type="spatial",
                 pooling_size=2, *, grid=None, kernel_size=None):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.proj_mpls = nn.Sequential(nn.Linear(self.in_dim, self.out_dim), nn.GELU(), nn.Linear(self.out_dim, self.out_dim))
        self.kernel_size = tuple(kernel_size) if kernel_size is not None else (1, 4, 4)
        self.num_patches_pre = [img // pch for img, pch in zip(image_size, patch_size)]
        self.num_patches_post = [self.num_patches_pre[i] // self.kernel_size[i] for i in range(len(self.num_patches_pre))]
        self.grid = tuple(grid) if grid is not None else (8, 16, 16)
        self.resolution_attention = resolution_attention_v3()

    def _params14(self):
        ra = self.resolution_attention
        return [self.proj_mpls[0].weight, self.proj_mpls[0].bias, self.proj_mpls[2].weight, self.proj_mpls[2].bias,
                ra.Wq.weight, ra.Wq.bias, ra.Wk.weight, ra.Wk.bias, ra.Wv.weight, ra.Wv.bias, ra.output_linear.weight,
                ra.output_linear.bias, ra.norm.weight, ra.norm.bias]

    def forward(self, visual_inputs):
        prec = _prec(self)
        batch = visual_inputs.size(0)
        if prec != PREC_F32 and not _unfused() and visual_inputs.dim() == 3 and visual_inputs.shape[-1] == self.in_dim:
            # 16-bit precisions: the whole packer as ONE autograd node over the fused driver (hs_packer_fwd / hs_packer_bwd)
            ps = self._params14()
            need_grad = torch.is_grad_enabled() and (visual_inputs.requires_grad or any(p.requires_grad for p in ps))
            meta = (self.grid, self.kernel_size, self.out_dim, prec, self.training, need_grad)
            return Fn.PackerFn.apply(visual_inputs, meta, *ps)
        HR = visual_inputs.contiguous().view(batch, *self.grid, 768)                    # :140
        LR = Fn.Pool3dFn.apply(HR, self.kernel_size)                                    # :141 avg_pool3d
        cd, ch, cw = (g // k for g, k in zip(self.grid, self.kernel_size))
        feats = self.resolution_attention(LR.view(batch, cd, ch, cw, 768), HR, self.kernel_size)
        h = Fn.LinearFn.apply(feats, self.proj_mpls[0].weight, self.proj_mpls[0].bias, prec)
        h = Fn.GeluFn.apply(h)
        h = Fn.LinearFn.apply(h, self.proj_mpls[2].weight, self.proj_mpls[2].bias, prec)
        return h.view(batch, -1, self.out_dim)

    @property
    def proj_out_num(self):
        num = 1
        for n in self.num_patches_post:
            num *= n
        return num


def build_mm_projector(config, delay_load=False, **kwargs):
    """multimodal_projector/builder.py:81-106 (only the shipped 'VisualPacker_3d_phi_v3' branch; the
    'baseline' branch of the reference names an un-imported class and cannot run there either)."""
    projector_type = getattr(config, "mm_projector_type")
    if projector_type == "VisualPacker_3d_phi_v3":
        return VisualPacker_3d_phi_v3(image_size=config.image_size, patch_size=config.patch_size,
                                      in_dim=config.mm_hidden_size, out_dim=config.hidden_size,
                                      layer_type=config.proj_layer_type, layer_num=config.proj_layer_num,
                                      pooling_type=config.proj_pooling_type, pooling_size=config.proj_pooling_size, **kwargs)
    raise ValueError(f"Unknown projector type: {projector_type}")


def encode_images(vision_tower, mm_projector, mm_projector2, images, images_2d):
    """LamedMetaForCausalLM.encode_images, dual-encoder branch (lamed_arch.py:122-132): two packers, concat on
    the token axis -> (B, 256, hidden).  ``mm_projector2=None`` reproduces the reference's bare-except fallback
    to the shared projector (:129-131)."""
    image_features = vision_tower(images, images_2d)
    if isinstance(image_features, tuple):
        f1, f2 = image_features
        a = mm_projector(f1)
        b = (mm_projector2 if mm_projector2 is not None else mm_projector)(f2)
        return torch.cat([a, b], dim=1)
    return mm_projector(image_features)


class LamedMetaForCausalLM:
    """The mixin form of the reference's LamedMetaForCausalLM (lamed_arch.py:114-155) for the methods on the vision path: a caller's
    model class lists this mixin where it listed the reference's and keeps ``get_model()`` (an object with ``get_vision_tower()``,
    ``mm_projector`` and, for the dual branch, ``mm_projector2``).  ``encode_images`` reproduces the reference's dispatch line by line in
    behaviour: the tuple test, the token-count tests on the LAST element (2048 = the native 8x16x16 patch grid -> two packers + cat;
    32 -> the legacy three-argument projector call), the bare-except fallback to the shared projector when ``mm_projector2`` is missing
    or fails (:129-131), the single-tensor branch.  Like the reference, a tuple with any other token count is returned UNCHANGED (the
    reference's caller then fails in its ``torch.cat``): geometries beyond the native one -- BASELINE configs[2]'s 256^3 volume has
    4096 patch tokens -- are admitted by listing their token count in ``dual_token_counts``."""

    dual_token_counts = (2048,)       # lamed_arch.py:125

    def get_model(self):
        raise NotImplementedError

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def encode_images(self, images, text_feature=None, images_2d=None):
        model = self.get_model()
        image_features = model.get_vision_tower()(images, images_2d)
        if isinstance(image_features, tuple):
            n = image_features[-1].shape[1]
            if n in self.dual_token_counts:
                ori, sec = image_features[0], image_features[1]
                ori = model.mm_projector(ori)
                try:
                    sec = model.mm_projector2(sec)
                except HsError:            # a HIP launch / shape / dtype failure is an error, not "no second projector"
                    raise
                except Exception:          # (the reference's bare except, :130: no second projector -> the shared one)
                    sec = model.mm_projector(sec)
                image_features = torch.cat([ori, sec], dim=1)
            elif n == 32:
                image_features = model.mm_projector(image_features[0], image_features[1], text_feature)
        else:
            image_features = model.mm_projector(image_features)
        return image_features

    def prepare_inputs_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images, images_2d):
        """lamed_arch.py:143-155: image tokens replace positions 1..n of the token embeddings."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        inputs_embeds = self.get_model().embed_tokens(input_ids)
        image_features = self.encode_images(images, inputs_embeds, images_2d)
        n = image_features.shape[1]
        inputs_embeds = torch.cat((inputs_embeds[:, :1, :], image_features.to(inputs_embeds.dtype), inputs_embeds[:, (n + 1):, :]), dim=1)
        return None, position_ids, attention_mask, past_key_values, inputs_embeds, labels
