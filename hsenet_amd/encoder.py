"""MI355X host mirror of the reference's LaMed 3D vision-encoder API
(Preprint/LaMed/src/model/multimodal_encoder/{vit.py,builder.py}).

Same class names, constructor signatures, forward signatures, return tuples, error behaviour and —
load-bearing for train_VLM.py:477-503's positional checkpoint copy — the same ``state_dict()`` keys IN THE
SAME ORDER (138 tensors for ViT_stage1, 150 for ViT_stage2).  The modules only *hold* parameters
(``nn.Linear`` / ``nn.LayerNorm`` objects are containers here, their ``forward`` is never called); all
arithmetic runs in libhsenet_hip.so through hsenet_amd.functional.  There is no CPU path: calling a module
with CPU tensors raises.

Dropout: MONAI blocks are built with dropout_rate=0.0 on this path (no-ops).  ``regular_attention`` hard-codes two
p=0.1 dropouts (on the attention probabilities and on the pre-residual output, vit.py:46-47,58,62) that fire whenever the
module is in train() — also for a "frozen" tower during MLLM finetuning (requires_grad_(False) does not call eval(),
lamed_arch.py:66).  Same here: in train() both sites drop with a counter-based Philox mask seeded from torch's default CPU
generator (``torch.manual_seed`` replays a step; the backward regenerates the mask); eval() is the deterministic function
the parity fixtures pin.  A frozen tower that must be deterministic needs ``.eval()`` — see DESIGN.md "Dropout".
"""
from __future__ import annotations

import os
from collections.abc import Sequence

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from ._lib import PREC_BF16, PREC_F16, PREC_F32


def _prec(module) -> int:
    p = getattr(module, "precision", None) or Fn.get_precision()
    return {"bf16": PREC_BF16, "f32": PREC_F32, "f16": PREC_F16}[p]


def _unfused() -> bool:
    """HS_MODULE_UNFUSED=1: the module API runs one autograd Function per reference operator (the form the exact-f32 parity mode and the
    'conv' patch embedding always use) instead of the fused embedding / packer drivers -- A/B runs and the equivalence tests."""
    return os.environ.get("HS_MODULE_UNFUSED") == "1"


class _Rearrange(nn.Module):
    """Parameter-free placeholder at index 0 of ``patch_embeddings`` so the Linear keeps MONAI's key
    ``patch_embeddings.1.*``; the rearrange itself is the hs_patchify kernel."""

    def __init__(self, patch_size):
        super().__init__()
        self.patch_size = tuple(patch_size)


class PatchEmbeddingBlock(nn.Module):
    """MONAI 1.3.0 PatchEmbeddingBlock (reference ctor call vit.py:428-437).  pos_embed='perceptron' (what every shipped call site
    passes: Rearrange + Linear, keys ``patch_embeddings.1.*``) or 'conv' (the constructors' DEFAULT, vit.py:237,375: a Conv3d with
    kernel = stride = patch, keys ``patch_embeddings.weight / .bias``).  A strided convolution over non-overlapping patches IS the
    perceptron form with the weight's axes in (c, p1, p2, p3) instead of (p1, p2, p3, c) order, so both run the same patchify + MFMA GEMM."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed="perceptron",
                 dropout_rate=0.0, spatial_dims=3):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        if pos_embed not in ("perceptron", "conv"):
            raise ValueError(f"pos_embed={pos_embed!r}: 'perceptron' or 'conv' (MONAI PatchEmbeddingBlock)")
        self.pos_embed = pos_embed
        if spatial_dims != 3:
            raise ValueError("only 3-D volumes are supported")
        img_size = tuple(img_size) if isinstance(img_size, Sequence) else (img_size,) * 3
        patch_size = tuple(patch_size) if isinstance(patch_size, Sequence) else (patch_size,) * 3
        for m, p in zip(img_size, patch_size):
            if m < p:
                raise ValueError("patch_size should be smaller than img_size.")
            if m % p != 0:
                raise ValueError("patch_size should be divisible by img_size for perceptron.")
        self.img_size, self.patch_size = img_size, patch_size
        self.n_patches = int(np.prod([i // p for i, p in zip(img_size, patch_size)]))
        self.patch_dim = int(in_channels * np.prod(patch_size))
        if pos_embed == "conv":        # (MONAI's _init_weights touches nn.Linear / nn.LayerNorm only: the conv keeps torch's default init)
            self.patch_embeddings = nn.Conv3d(in_channels, hidden_size, kernel_size=patch_size, stride=patch_size)
        else:
            self.patch_embeddings = nn.Sequential(_Rearrange(patch_size), nn.Linear(self.patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, hidden_size))
        self.dropout = nn.Dropout(dropout_rate)
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        if pos_embed == "perceptron":
            nn.init.trunc_normal_(self.patch_embeddings[1].weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
            nn.init.zeros_(self.patch_embeddings[1].bias)

    def fused_weight_bias(self):
        """(weight, bias) PARAMETERS whose memory the fused drivers can read as the [hidden, p1*p2*p3*c] projection of the patchified
        volume, or None: the perceptron Linear always; the Conv3d when in_channels == 1 -- its [hidden, 1, p1, p2, p3] weight IS that
        matrix in memory (with more channels the two orders differ: (c, p1, p2, p3) against (p1, p2, p3, c))."""
        if self.pos_embed == "perceptron":
            return self.patch_embeddings[1].weight, self.patch_embeddings[1].bias
        if self.patch_embeddings.weight.shape[1] == 1:
            return self.patch_embeddings.weight, self.patch_embeddings.bias
        return None

    def linear_weight_bias(self):
        """(weight [hidden, p1*p2*p3*c], bias) of the patch projection in the rearranged feature order '(p1 p2 p3 c)' the patchify
        kernel emits; for 'conv' a (differentiable) re-ordering of the Conv3d weight [hidden, c, p1, p2, p3]."""
        if self.pos_embed == "conv":
            w = self.patch_embeddings.weight
            return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1), self.patch_embeddings.bias
        lin = self.patch_embeddings[1]
        return lin.weight, lin.bias


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)


class SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.num_heads = num_heads


class TransformerBlock(nn.Module):
    """MONAI TransformerBlock parameter container; registration order mlp, norm1, attn, norm2."""

    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if dropout_rate != 0.0:
            raise ValueError("dropout_rate must be 0.0 (the reference builds its blocks with 0.0, CLIP_stage1.py:30)")
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate, qkv_bias, save_attn)
        self.norm2 = nn.LayerNorm(hidden_size)

    def ordered_params(self):
        """the 11 tensors every block has, in MONAI's state-dict order (the drivers' layer_params); a qkv_bias=True block's
        ``attn.qkv.bias`` travels separately (hs_vit_desc.qkv_bias / .qkv_bias_grad)."""
        return [self.mlp.linear1.weight, self.mlp.linear1.bias, self.mlp.linear2.weight, self.mlp.linear2.bias,
                self.norm1.weight, self.norm1.bias, self.attn.out_proj.weight, self.attn.out_proj.bias,
                self.attn.qkv.weight, self.norm2.weight, self.norm2.bias]


class regular_attention(nn.Module):
    """vit.py:36-64 — single-head cross-attention used as the slice-guided scorer of the 2E3 encoder."""

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
        if mask is not None:
            raise ValueError("mask is never passed on the HSENet path (vit.py:335)")
        prec = _prec(self)
        q = Fn.LinearFn.apply(Query, self.Wq.weight, self.Wq.bias, prec)
        k = Fn.LinearFn.apply(Key, self.Wk.weight, self.Wk.bias, prec)
        v = Fn.LinearFn.apply(Value, self.Wv.weight, self.Wv.bias, prec)
        # The reference hard-codes Dropout(0.1) on the probabilities and on the pre-residual output (vit.py:46-47, 58, 62) and
        # they fire whenever the module is in train() -- also for the "frozen" tower during MLLM finetuning.  Same here:
        # Philox masks keyed by torch's generator (torch.manual_seed replays them); eval() gives the deterministic function.
        train = self.training
        pa, po = (self.dropout.p, self.dropout_2.p) if train else (0.0, 0.0)
        x, self.attn = Fn.CrossAttnFn.apply(q, k, v, pa, *(Fn.draw_dropout_seed() if pa > 0 else (0, 0)))
        x = Fn.LinearFn.apply(x, self.output_linear.weight, self.output_linear.bias, prec)
        if po > 0:
            x = Fn.DropoutFn.apply(x, po, *Fn.draw_dropout_seed())
        x = Fn.LayerNormFn.apply(q, x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x, self.attn


class _ViTBase(nn.Module):
    def _build(self, in_channels, img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads, pos_embed,
               classification, dropout_rate, spatial_dims, qkv_bias, save_attn):
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if hidden_size // num_heads != 64:
            raise ValueError("the MI355X attention kernels are specialised for head_dim 64 (ViT-B 768/12, ViT-S 384/6)")
        self.hidden_size = hidden_size
        self.mlp_dim = mlp_dim
        self.num_heads = num_heads
        self.classification = classification
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed,
                                                   dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate, qkv_bias, save_attn)
                                     for _ in range(num_layers)])

    def _embed(self, x, prec):
        pe = self.patch_embedding
        if tuple(x.shape[2:]) != pe.img_size:
            raise ValueError(f"input volume {tuple(x.shape[2:])} does not match img_size {pe.img_size}")
        w, b = pe.linear_weight_bias()
        return Fn.PatchEmbedFn.apply(x, w, b, pe.patch_size, prec)

    def _fused_ok(self, prec, *inputs) -> bool:
        """the fused embedding driver (one autograd node for P1 + token assembly + the 2E3 scorer) serves the 16-bit precisions of the
        shipped 'perceptron' embedding and of the constructors' default 'conv' embedding on one-channel volumes (fused_weight_bias);
        inputs that ask for a gradient (never the case for volumes / slice features) take the per-operator Functions."""
        return (prec != PREC_F32 and self.patch_embedding.fused_weight_bias() is not None and not _unfused()
                and not any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs))

    def _head16(self):
        """the embedding driver's 16 parameter slots (hs_embed_*): [cls, pos, pe.w, pe.b, score.w, score.b, Wq, Wk, Wv, Wo (w, b each), norm.w, norm.b]."""
        pe = self.patch_embedding
        head = [getattr(self, "cls_token", None), pe.position_embeddings, *pe.fused_weight_bias()]
        sga = getattr(self, "slice_guided_attention", None)
        if sga is not None:
            head += [self.patch_score_proj.weight, self.patch_score_proj.bias]
            for lin in (sga.Wq, sga.Wk, sga.Wv, sga.output_linear):
                head += [lin.weight, lin.bias]
            head += [sga.norm.weight, sga.norm.bias]
        return head + [None] * (16 - len(head))

    def _run_fused(self, x, image_2d, prec):
        """this encoder alone through the fused node chain (hsenet_amd/fused.py)."""
        from . import fused
        (y, hidden), = fused.run_encoders([fused.EncJob(self, x, image_2d, prec)])
        return y, hidden

    def qkv_biases(self):
        """[attn.qkv.bias per block] for qkv_bias=True encoders, [] otherwise."""
        return [blk.attn.qkv.bias for blk in self.blocks] if len(self.blocks) and self.blocks[0].attn.qkv.bias is not None else []

    def _stack(self, tokens, prec):
        qb = self.qkv_biases()
        params = [p for blk in self.blocks for p in blk.ordered_params()] + [self.norm.weight, self.norm.bias] + qb
        need_grad = torch.is_grad_enabled() and (tokens.requires_grad or any(p.requires_grad for p in params))
        cache = self.__dict__.get("_w16cache")
        if cache is None:
            cache = self.__dict__["_w16cache"] = Fn.W16Cache()
        y, hidden = Fn.BlocksFn.apply(tokens, self.num_heads, self.mlp_dim, prec, need_grad, cache, len(qb), *params)
        return y, [hidden[i] for i in range(hidden.shape[0])]


class ViT_stage1(_ViTBase):
    """vit.py:360-469 — the 3D-ViT encoder.  forward(x) -> (tokens [B,N,d] after `norm`, [hidden states])."""

    def __init__(self, in_channels: int, img_size, patch_size, hidden_size: int = 768, mlp_dim: int = 3072,
                 num_layers: int = 12, num_heads: int = 12, pos_embed: str = "conv", classification: bool = False,
                 num_classes: int = 2, dropout_rate: float = 0.0, spatial_dims: int = 3, post_activation="Tanh",
                 qkv_bias: bool = False, save_attn: bool = False) -> None:
        super().__init__()
        self._build(in_channels, img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads, pos_embed,
                    classification, dropout_rate, spatial_dims, qkv_bias, save_attn)
        self.norm = nn.LayerNorm(hidden_size)
        if self.classification:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))

    def forward(self, x, k=None, visual_encoder_2D=None, text_features=None, image_path=None):
        prec = _prec(self)
        if self._fused_ok(prec, x):
            return self._run_fused(x, None, prec)
        pe = self._embed(x, prec)
        cls = self.cls_token if hasattr(self, "cls_token") else None
        tokens, _ = Fn.AssembleFn.apply(pe, self.patch_embedding.position_embeddings, cls, None)
        return self._stack(tokens, prec)


class ViT_stage2(_ViTBase):
    """vit.py:222-357 — the 2D-enhanced "2E3" encoder: patch tokens are re-weighted by a slice-guided score
    before the block stack.  forward(x, image_2d) -> (tokens, [hidden states])."""

    def __init__(self, in_channels: int, img_size, patch_size, hidden_size: int = 768, mlp_dim: int = 3072,
                 num_layers: int = 12, num_heads: int = 12, pos_embed: str = "conv", classification: bool = False,
                 num_classes: int = 2, dropout_rate: float = 0.0, spatial_dims: int = 3, post_activation="Tanh",
                 qkv_bias: bool = False, save_attn: bool = False) -> None:
        super().__init__()
        self._build(in_channels, img_size, patch_size, hidden_size, mlp_dim, num_layers, num_heads, pos_embed,
                    classification, dropout_rate, spatial_dims, qkv_bias, save_attn)
        self.patch_score_proj = nn.Linear(hidden_size, 1)
        self.patch_score_norm = nn.Sigmoid()
        self.slice_guided_attention = regular_attention()
        self.norm = nn.LayerNorm(hidden_size)
        if self.classification:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))

    def forward(self, x, image_2d, k=None, visual_encoder_2D=None, text_features=None, image_path=None):
        prec = _prec(self)
        batch = x.size(0)
        if self._fused_ok(prec, x, image_2d):
            self.slice_guided_attention.attn = None       # (the attention map stays inside the fused driver's workspace)
            return self._run_fused(x, image_2d, prec)
        pe = self._embed(x, prec)
        pos = self.patch_embedding.position_embeddings
        cls = self.cls_token if hasattr(self, "cls_token") else None
        xe = _PosAdd.apply(pe, pos)                   # x = patch_embedding(x): embedding + position (vit.py:325)
        sem = image_2d.to(x.device).view(batch, 32, -1)                                   # vit.py:332
        patch_score, _ = self.slice_guided_attention(xe, sem, sem)                         # vit.py:335
        score = Fn.RowdotSigmoidFn.apply(patch_score, self.patch_score_proj.weight, self.patch_score_proj.bias)  # :338-339
        tokens, _ = Fn.AssembleFn.apply(pe, pos, cls, score)                               # :345-349
        return self._stack(tokens, prec)


class _PosAdd(torch.autograd.Function):
    """pe + position_embeddings as a differentiable op (the scorer's Query input, vit.py:325,335)."""

    @staticmethod
    def forward(ctx, pe, pos):
        B, Np, d = pe.shape
        out = torch.empty_like(pe)
        Fn.L.call("hs_assemble_tokens", pe.contiguous().data_ptr(), pos.data_ptr(), None, None, out.data_ptr(), None, B, Np, d, Fn._s())
        ctx.shape = pos.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        B, Np, d = dout.shape
        dpos = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        Fn.L.call("hs_assemble_tokens_bwd", dout.data_ptr(), None, None, None, dpos.data_ptr(), None, None, B, Np, d, Fn._s())
        return dout, dpos


class ViT3DTower_dual_encoders(nn.Module):
    """vit.py:891-960 — runs the 3D-ViT then the 2E3-ViT on the same volume."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.select_layer = config.vision_select_layer
        self.select_feature = config.vision_select_feature
        self.remain_2d3d_ViT_type = getattr(config, "remain_2d3d_ViT_type", "dual_vits")
        kw = dict(in_channels=self.config.image_channel, img_size=self.config.image_size, patch_size=self.config.patch_size,
                  pos_embed="perceptron", spatial_dims=len(self.config.patch_size), classification=True)
        self.vision_tower_stage1 = ViT_stage1(**kw)
        self.vision_tower_stage2 = ViT_stage2(**kw)

    # two_streams: None = the engines' rule (the 2E3 encoder's kernels on a side HIP stream below 16 volumes per GPU, joined inside every
    # autograd node: see hsenet_amd/fused.py), True / False force it
    two_streams = None

    def _encode_both(self, images, images_2d):
        """both encoders as ONE chain of fused autograd nodes (16-bit precisions; hsenet_amd/fused.py) -- or, when that form does not
        apply (exact-f32 mode, 'conv' embedding, forward hooks on the encoders, inputs that want gradients), one after the other as
        the reference writes it (vit.py:928-929)."""
        s1, s2 = self.vision_tower_stage1, self.vision_tower_stage2
        prec = _prec(s1)
        hooked = any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in (s1, s2))
        if (not hooked and prec == _prec(s2) and s1._fused_ok(prec, images) and s2._fused_ok(prec, images, images_2d)
                and len(s1.blocks) == len(s2.blocks)):
            from . import fused
            with Fn.shared_patches():          # both encoders embed the same volume: it is patchified once
                jobs = [fused.EncJob(s1, images, None, prec), fused.EncJob(s2, images, images_2d, prec)]
            s2.slice_guided_attention.attn = None
            ts = self.two_streams if self.two_streams is not None else fused.dual_streams_default(images.shape[0])
            (f1, _), (f2, _) = fused.run_encoders(jobs, two_streams=ts)
            return f1, f2
        with Fn.shared_patches():
            f1, _ = s1(images)
            f2, _ = s2(images, images_2d)
        return f1, f2

    def forward(self, images, images_2d):
        f1, f2 = self._encode_both(images, images_2d)
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
    def dtype(self):
        return self.vision_tower_stage1.norm.weight.dtype

    @property
    def device(self):
        return self.vision_tower_stage1.norm.weight.device

    @property
    def hidden_size(self):
        return self.vision_tower_stage1.hidden_size


def build_vision_tower(config, **kwargs):
    """multimodal_encoder/builder.py:4-11.  The legacy single-encoder 'vit3d' towers are ablations outside the
    hot path and are not provided."""
    vision_tower = getattr(config, "vision_tower", None)
    if "vit_stage2_dual_encoders" in vision_tower.lower():
        return ViT3DTower_dual_encoders(config, **kwargs)
    elif "vit3d" in vision_tower.lower():
        raise ValueError("vision tower 'vit3d' (legacy single-encoder ablation) is outside the MI355X hot path; "
                         "use 'vit_stage2_dual_encoders'")
    else:
        raise ValueError(f"Unknown vision tower: {vision_tower}")
