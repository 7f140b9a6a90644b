"""CPU oracle for the HSENet dual 3D vision-encoder hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain-PyTorch fp32 restatement of the reference algorithm.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``hsenet_amd/`` imports it, and the product path has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * ``regular_attention``, ``ViT_stage1``, ``ViT_stage2``, ``ViT3DTower_dual_encoders``,
    ``resolution_attention_v3``, ``VisualPacker_3d_phi_v3``: PINNED — ``oracle/make_goldens.py``
    imports the reference modules from /root/reference in the build container, loads identical
    seeded weights and asserts equality (<=1e-6), then writes ``tests/golden/*.npz``.
  * ``PatchEmbeddingBlock`` / ``TransformerBlock`` / ``SABlock`` / ``MLPBlock``: the reference takes
    these from third-party MONAI 1.3.0 (requirements.txt:62), which is neither vendored nor
    installed and cannot be fetched.  They are restated from MONAI's published semantics
     -> **parity unpinned** against MONAI itself.  The arithmetic is
    cross-checked against two independent implementations that ARE present here
    (``torch.nn.MultiheadAttention`` and HuggingFace ``ViTLayer``) in tests/test_oracle.py.

Reference citations are relative to /root/reference/Preprint/LaMed/src/model/.
"""
from __future__ import annotations

import math
from contextlib import contextmanager
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from einops import rearrange

# --------------------------------------------------------------------------------------
# optional bf16 emulation of the reference's autocast policy ("Precision"):
# Linear / matmul take bf16-rounded inputs with fp32 accumulation; everything else fp32.
# Used only to *predict* the perf-mode error on CPU; the parity oracle is plain fp32.
# --------------------------------------------------------------------------------------
_EMULATE_BF16 = False


@contextmanager
def emulate_bf16(enabled: bool = True):
    global _EMULATE_BF16
    old = _EMULATE_BF16
    _EMULATE_BF16 = enabled
    try:
        yield
    finally:
        _EMULATE_BF16 = old


def _r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if _EMULATE_BF16 else x


def _linear(x, weight, bias=None):
    return F.linear(_r(x), _r(weight), bias)


def _matmul(a, b):
    return torch.matmul(_r(a), _r(b))


# --------------------------------------------------------------------------------------
# MONAI 1.3.0 blocks (restated).  Call sites: multimodal_encoder/vit.py:428-443
# --------------------------------------------------------------------------------------
# MONAI PatchEmbeddingBlock's 'perceptron' rearrangement.  The one in-tree witness of MONAI's layout is the repr the reference's
# authors pasted as a comment behind every PatchEmbeddingBlock call (vit.py:144, 299, 437, 508, 788): oracle/make_goldens.py reads those
# comments and asserts that each of them carries exactly this string (check_rearrange_witness).
REARRANGE_PATTERN = "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)"


class _Rearrange(nn.Module):
    """Stand-in for einops.layers.torch.Rearrange at index 0 of ``patch_embeddings`` (no params)."""

    def __init__(self, patch_size):
        super().__init__()
        self.patch_size = tuple(patch_size)

    def forward(self, x):
        p1, p2, p3 = self.patch_size
        # pattern recorded in the reference at vit.py:437
        return rearrange(x, REARRANGE_PATTERN, p1=p1, p2=p2, p3=p3)


class PatchEmbeddingBlock(nn.Module):
    """MONAI PatchEmbeddingBlock (ctor call vit.py:428-437, use :455): pos_embed='perceptron' (Rearrange + Linear: every shipped call
    site) or 'conv' (the constructors' default, vit.py:237,375: Conv3d with kernel = stride = patch, then flatten(2).transpose(-1, -2),
    MONAI 1.3.0 patchembedding.py forward)."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads,
                 pos_embed="perceptron", dropout_rate=0.0, spatial_dims=3):
        super().__init__()
        if pos_embed not in ("perceptron", "conv"):
            raise ValueError("pos_embed: 'perceptron' or 'conv'")
        self.pos_embed = pos_embed
        img_size = tuple(img_size)
        patch_size = tuple(patch_size)
        self.n_patches = int(np.prod([i // p for i, p in zip(img_size, patch_size)]))
        self.patch_dim = int(in_channels * np.prod(patch_size))
        if pos_embed == "conv":
            self.patch_embeddings = nn.Conv3d(in_channels, hidden_size, kernel_size=patch_size, stride=patch_size)
        else:
            self.patch_embeddings = nn.Sequential(_Rearrange(patch_size), nn.Linear(self.patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, hidden_size))
        self.dropout = nn.Dropout(dropout_rate)
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)

    def forward(self, x):
        if self.pos_embed == "conv":
            x = self.patch_embeddings(x).flatten(2).transpose(-1, -2)
            return self.dropout(x + self.position_embeddings)
        t = self.patch_embeddings[0](x)
        lin = self.patch_embeddings[1]
        x = _linear(t, lin.weight, lin.bias)
        return self.dropout(x + self.position_embeddings)


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)

    def forward(self, x):
        x = _linear(x, self.linear1.weight, self.linear1.bias)
        x = F.gelu(x)  # exact erf GELU (MONAI act="GELU")
        return _linear(x, self.linear2.weight, self.linear2.bias)


class SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5

    def forward(self, x):
        qkv = _linear(x, self.qkv.weight, self.qkv.bias)
        # MONAI: Rearrange('b h (qkv l d) -> qkv b l h d', qkv=3, l=num_heads)
        q, k, v = rearrange(qkv, "b n (qkv l d) -> qkv b l n d", qkv=3, l=self.num_heads)
        att = (_matmul(q, k.transpose(-1, -2)) * self.scale).softmax(dim=-1)
        y = _matmul(att, v)
        y = rearrange(y, "b l n d -> b n (l d)")
        return _linear(y, self.out_proj.weight, self.out_proj.bias)


class TransformerBlock(nn.Module):
    """MONAI TransformerBlock (positional ctor call at vit.py:440). Registration order matters."""

    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate, qkv_bias, save_attn)
        self.norm2 = nn.LayerNorm(hidden_size)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


# --------------------------------------------------------------------------------------
# multimodal_encoder/vit.py restated
# --------------------------------------------------------------------------------------
def attention(query, key, value, mask=None, dropout=None):
    """vit.py:25-33 and spatial_pooling_projector.py:8-16 (identical)."""
    d_k = query.size(-1)
    scores = _matmul(query, key.transpose(-2, -1)) / math.sqrt(d_k)
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e9)
    p_attn = F.softmax(scores, dim=-1)
    if dropout is not None:
        p_attn = dropout(p_attn)
    return _matmul(p_attn, value), p_attn


class regular_attention(nn.Module):
    """vit.py:36-64."""

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

    def forward(self, Query, Key, Value, context=None, mask=None):
        q = _linear(Query, self.Wq.weight, self.Wq.bias)
        k = _linear(Key, self.Wk.weight, self.Wk.bias)
        v = _linear(Value, self.Wv.weight, self.Wv.bias)
        x, self.attn = attention(q, k, v, mask=mask, dropout=self.dropout)
        x = _linear(x, self.output_linear.weight, self.output_linear.bias)
        x = self.norm(q + self.dropout_2(x))
        return x, self.attn


def _check_vit_args(dropout_rate, hidden_size, num_heads):
    if not (0 <= dropout_rate <= 1):
        raise ValueError("dropout_rate should be between 0 and 1.")
    if hidden_size % num_heads != 0:
        raise ValueError("hidden_size should be divisible by num_heads.")


class ViT_stage1(nn.Module):
    """vit.py:360-469."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size=768, mlp_dim=3072, num_layers=12,
                 num_heads=12, pos_embed="conv", classification=False, num_classes=2, dropout_rate=0.0,
                 spatial_dims=3, post_activation="Tanh", qkv_bias=False, save_attn=False):
        super().__init__()
        _check_vit_args(dropout_rate, hidden_size, num_heads)
        self.hidden_size = hidden_size
        self.classification = classification
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads,
                                                   pos_embed, dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate, qkv_bias,
                                                      save_attn) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)
        if self.classification:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))

    def forward(self, x, k=None, visual_encoder_2D=None, text_features=None, image_path=None):
        x = self.patch_embedding(x.clone())
        if hasattr(self, "cls_token"):
            cls = self.cls_token.expand(x.shape[0], -1, -1)
            x = torch.cat((cls, x), dim=1)
        hidden = []
        for blk in self.blocks:
            x = blk(x)
            hidden.append(x)
        x = self.norm(x)
        return x, hidden


class ViT_stage2(nn.Module):
    """vit.py:222-357 (the "2E3" encoder)."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size=768, mlp_dim=3072, num_layers=12,
                 num_heads=12, pos_embed="conv", classification=False, num_classes=2, dropout_rate=0.0,
                 spatial_dims=3, post_activation="Tanh", qkv_bias=False, save_attn=False):
        super().__init__()
        _check_vit_args(dropout_rate, hidden_size, num_heads)
        self.hidden_size = hidden_size
        self.classification = classification
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads,
                                                   pos_embed, dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate, qkv_bias,
                                                      save_attn) for _ in range(num_layers)])
        self.patch_score_proj = nn.Linear(hidden_size, 1)
        self.patch_score_norm = nn.Sigmoid()
        self.slice_guided_attention = regular_attention()
        self.norm = nn.LayerNorm(hidden_size)
        if self.classification:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))

    def forward(self, x, image_2d, k=None, visual_encoder_2D=None, text_features=None, image_path=None):
        batch = x.size(0)
        x = self.patch_embedding(x.clone())
        batch_size, num_patches, slice_num = x.size(0), x.size(1), 32
        sem = image_2d.to(x.device).view(batch, slice_num, -1)                      # vit.py:332
        patch_score, _ = self.slice_guided_attention(x, sem, sem)                    # vit.py:335
        patch_score = _linear(patch_score, self.patch_score_proj.weight, self.patch_score_proj.bias)
        scores_flat = self.patch_score_norm(patch_score.view(batch_size, num_patches))
        xw = x * scores_flat.unsqueeze(-1)                                           # vit.py:345
        if hasattr(self, "cls_token"):
            cls = self.cls_token.expand(xw.shape[0], -1, -1)
            xw = torch.cat((cls, xw), dim=1)
        hidden = []
        for blk in self.blocks:
            xw = blk(xw)
            hidden.append(xw)
        xw = self.norm(xw)
        return xw, hidden


class ViT3DTower_dual_encoders(nn.Module):
    """vit.py:891-960."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.select_layer = config.vision_select_layer
        self.select_feature = config.vision_select_feature
        self.remain_2d3d_ViT_type = getattr(config, "remain_2d3d_ViT_type", "dual_vits")
        kw = dict(in_channels=config.image_channel, img_size=config.image_size, patch_size=config.patch_size,
                  pos_embed="perceptron", spatial_dims=len(config.patch_size), classification=True)
        self.vision_tower_stage1 = ViT_stage1(**kw)
        self.vision_tower_stage2 = ViT_stage2(**kw)

    def forward(self, images, images_2d):
        f1, _ = self.vision_tower_stage1(images)
        f2, _ = self.vision_tower_stage2(images, images_2d)
        if self.select_feature == "patch":
            f1, f2 = f1[:, 1:], f2[:, 1:]
        elif self.select_feature == "cls_patch":
            pass
        else:
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        if self.remain_2d3d_ViT_type == "dual_vits":
            return f1, f2
        elif self.remain_2d3d_ViT_type == "3d_vit":
            return f1
        elif self.remain_2d3d_ViT_type == "2e3_vit":
            return f2

    @property
    def hidden_size(self):
        return self.vision_tower_stage1.hidden_size


# --------------------------------------------------------------------------------------
# multimodal_projector/spatial_pooling_projector.py restated
# --------------------------------------------------------------------------------------
class resolution_attention_v3(nn.Module):
    """spatial_pooling_projector.py:48-83."""

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
        S_d, S_w, S_h = kernel_size
        sd, sw, sh = HR_image.size(1), HR_image.size(2), HR_image.size(3)
        batch, dim = LR_image.size(0), LR_image.size(-1)
        LR = LR_image.contiguous().view(batch, -1, dim).unsqueeze(1).permute(0, 2, 1, 3)     # (B, cells, 1, D)
        HR = HR_image.reshape(batch, sd // S_d, S_d, sw // S_w, S_w, sh // S_h, S_h, dim)
        HR = HR.permute(0, 2, 4, 6, 1, 3, 5, 7).contiguous()
        HR = HR.view(batch, S_d * S_w * S_h, (sd // S_d) * (sw // S_w) * (sh // S_h), dim).permute(0, 2, 1, 3)
        q = _linear(LR, self.Wq.weight, self.Wq.bias)
        k = _linear(HR, self.Wk.weight, self.Wk.bias)
        v = _linear(HR, self.Wv.weight, self.Wv.bias)
        x, self.attn = attention(q, k, v, mask=mask, dropout=self.dropout)
        x = x.reshape(batch, -1, dim)
        q = q.reshape(batch, -1, dim)
        x = _linear(x, self.output_linear.weight, self.output_linear.bias)
        return self.norm(q + self.dropout_2(x))


class VisualPacker_3d_phi_v3(nn.Module):
    """spatial_pooling_projector.py:121-153.

    ``grid``/``kernel_size`` default to the reference's hard-coded (8,16,16)/(1,4,4) (:132,:140).  They are
    exposed (keyword-only extension) because BASELINE configs 2-5 use a 16x16x16 token grid, for which the
    reference-sanctioned kernel is (2,4,4) (:166-167) -> still 128 tokens per packer.
    """

    def __init__(self, image_size, patch_size, in_dim, out_dim, layer_type, layer_num, pooling_type="spatial",
                 pooling_size=2, *, grid=None, kernel_size=None):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.proj_mpls = nn.Sequential(nn.Linear(in_dim, out_dim), nn.GELU(), nn.Linear(out_dim, out_dim))
        self.kernel_size = tuple(kernel_size) if kernel_size is not None else (1, 4, 4)
        self.num_patches_pre = [img // pch for img, pch in zip(image_size, patch_size)]
        self.num_patches_post = [self.num_patches_pre[i] // self.kernel_size[i] for i in range(3)]
        self.grid = tuple(grid) if grid is not None else (8, 16, 16)
        self.resolution_attention = resolution_attention_v3()

    def forward(self, visual_inputs):
        batch = visual_inputs.size(0)
        HR = visual_inputs.view(batch, *self.grid, 768)
        LR = F.avg_pool3d(HR.clone().permute(0, 4, 1, 2, 3), kernel_size=self.kernel_size).permute(0, 2, 3, 4, 1)
        feats = self.resolution_attention(LR, HR, self.kernel_size)
        h = _linear(feats, self.proj_mpls[0].weight, self.proj_mpls[0].bias)
        h = F.gelu(h)
        h = _linear(h, self.proj_mpls[2].weight, self.proj_mpls[2].bias)
        return h.view(batch, -1, self.out_dim)

    @property
    def proj_out_num(self):
        num = 1
        for n in self.num_patches_post:
            num *= n
        return num


def encode_images(tower, mm_projector, mm_projector2, images, images_2d):
    """lamed_arch.py:122-141, the dual-encoder branch (tuple, two packers, cat on dim 1)."""
    f1, f2 = tower(images, images_2d)
    a = mm_projector(f1)
    b = (mm_projector2 if mm_projector2 is not None else mm_projector)(f2)
    return torch.cat([a, b], dim=1)


def clip_encode_image(vit, mm_vision_proj, image, image_2d=None):
    """CLIP_stage1.py:97-102 / CLIP_stage2.py:105-110: proj on all tokens, normalise; caller keeps [:,0]."""
    feats, _ = vit(image) if image_2d is None else vit(image, image_2d)
    feats = _linear(feats, mm_vision_proj.weight, mm_vision_proj.bias)
    return F.normalize(feats, dim=-1)


def clip_contrastive_loss(image_features, text_features, logit_scale):
    """CLIP_stage1.py:141-155 at world size 1 (gather_features is the identity there): the raw
    ``logit_scale`` parameter is a plain multiplier (not exponentiated, :148)."""
    logits_per_image = logit_scale * image_features @ text_features.t()
    logits_per_text = logits_per_image.t()
    labels = torch.arange(image_features.shape[0], device=image_features.device)
    return (F.cross_entropy(logits_per_image, labels) + F.cross_entropy(logits_per_text, labels)) / 2


# --------------------------------------------------------------------------------------
# seeded synthetic parameters / inputs (shared recipe: both the oracle and the HIP modules
# are filled through load_state_dict from this, so neither side ships weights)
# --------------------------------------------------------------------------------------
def seeded_state_dict(module: nn.Module, seed: int) -> dict:
    """Every tensor non-trivial so that a swapped/ignored parameter is visible:
    weights ~ N(0, 0.02) (x2.5 for 1-D 'weight' of LayerNorm around 1), biases ~ N(0, 0.02)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, t in module.state_dict().items():
        a = rng.standard_normal(tuple(t.shape), dtype=np.float32) * 0.02
        if name.endswith("weight") and t.dim() == 1:       # LayerNorm gamma
            a = 1.0 + 2.5 * a
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32))     # (asarray: a 0-d parameter such as logit_scale draws a scalar)
    return out


def synthetic_volume(shape: Sequence[int], seed: int) -> torch.Tensor:
    """U[0,1) fp32 — the dataset contract 'normalized 0-1' (dataset/multi_dataset.py:246)."""
    return torch.from_numpy(np.random.default_rng(seed).random(tuple(shape), dtype=np.float32))


def synthetic_slices(batch: int, seed: int) -> torch.Tensor:
    """(B,32,768) ~ N(0,1): stand-in for the pre-computed BiomedCLIP slice features."""
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((batch, 32, 768), dtype=np.float32))


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """The parity metric: ||a-b||_F / ||b||_F (SURVEY.md section 7 'hard parts')."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
