"""The 2-D branch that produces ``image_2d`` for the 2E3 encoder.

Reference: every CT volume is resized to 32 x 224 x 224 (trilinear), each of the 32 slices is fed as a 3-channel image to
the BiomedCLIP vision trunk, and the 32 pooled 768-vectors are what ``ViT_stage2.forward(x, image_2d)`` receives
(multimodal_encoder/vit.py:515-518, 525-533 for the on-the-fly form; Data/data_processing/CT-RATE/CT-RATE_2D_to_npy_file.py:
80-94 for the pre-computed ``.npy`` form: ``model.visual.trunk(batch)``).

``model.visual.trunk`` is timm's ``vit_base_patch16_224`` with ``num_classes=0`` (third-party: open_clip / timm, both absent
here, weights not available offline) -> its published architecture is restated: Conv2d(3,768,16,16) patch embedding, cls
token, 197 learned positions (added to the cls token too), 12 pre-LN blocks with BIASED qkv, LayerNorm eps 1e-6, exact
GELU, final LayerNorm, cls-token pooling.  Parity for this row is therefore "unpinned" against the reference and pinned
instead (on the CPU restatement, in the CPU test suite) against HuggingFace's ViT implementation of the same architecture.

`BiomedClipTrunk` keeps timm's state-dict keys and order, so the released BiomedCLIP trunk weights load with
``load_state_dict``.  Forward only (the reference freezes the trunk, vit.py:518); runs through the fused block driver with
``hs_vit_desc.ln_eps = 1e-6`` and ``hs_vit_desc.qkv_bias``."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L
from . import functional as Fn
from . import preprocess as P


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.qkv = nn.Linear(d, 3 * d, bias=True)
        self.proj = nn.Linear(d, d)


class _Mlp(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.fc1 = nn.Linear(d, hidden)
        self.fc2 = nn.Linear(hidden, d)


class _Block(nn.Module):                     # timm.models.vision_transformer.Block registration order
    def __init__(self, d, hidden):
        super().__init__()
        self.norm1 = nn.LayerNorm(d, eps=1e-6)
        self.attn = _Attn(d)
        self.norm2 = nn.LayerNorm(d, eps=1e-6)
        self.mlp = _Mlp(d, hidden)

    def driver_params(self):                 # the fused driver's per-layer order (MONAI's): W1 b1 W2 b2 n1 Wo bo Wqkv n2
        return [self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, self.norm1.weight, self.norm1.bias,
                self.attn.proj.weight, self.attn.proj.bias, self.attn.qkv.weight, self.norm2.weight, self.norm2.bias]


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, d, patch):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, d, kernel_size=patch, stride=patch)


class BiomedClipTrunk(nn.Module):
    """timm ``VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=0)``."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0):
        super().__init__()
        if embed_dim // num_heads != 64 or embed_dim % num_heads:
            raise ValueError("the MI355X attention kernels are specialised for head_dim 64")
        if img_size % patch_size:
            raise ValueError("img_size must be a multiple of patch_size")
        self.img_size, self.patch_size, self.in_chans = img_size, patch_size, in_chans
        self.embed_dim, self.num_heads, self.mlp_dim = embed_dim, num_heads, int(embed_dim * mlp_ratio)
        self.num_patches = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.num_patches + 1, embed_dim) * 0.02)
        self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch_size)
        self.blocks = nn.Sequential(*[_Block(embed_dim, self.mlp_dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.requires_grad_(False)            # vit.py:518

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [n, 3, 224, 224] fp32 on the GPU -> pooled cls features [n, 768]."""
        Fn._need(x)
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.in_chans, self.img_size, self.img_size):
            raise ValueError(f"expected [n, {self.in_chans}, {self.img_size}, {self.img_size}], got {tuple(x.shape)}")
        prec = Fn._PREC[Fn.get_precision()]
        n, p, d = x.shape[0], self.patch_size, self.embed_dim
        # Conv2d(k = stride = 16) == Linear on patches.  hs_patchify orders a patch's features (ph, pw, c) with c fastest;
        # the conv weight is [d][c][ph][pw] -> permute once per call (768 x 768: parameter plumbing, not data path)
        w = self.patch_embed.proj.weight.permute(0, 2, 3, 1).reshape(d, -1).contiguous()
        pe = Fn.PatchEmbedFn.apply(x.unsqueeze(2), w, self.patch_embed.proj.bias, (1, p, p), prec)       # [n, 196, d]
        # timm adds the position embedding to the cls token as well: fold it into the cls row handed to the assembler
        cls = (self.cls_token + self.pos_embed[:, :1]).contiguous()
        tokens, _ = Fn.AssembleFn.apply(pe, self.pos_embed[:, 1:].contiguous(), cls, None)
        params = [q for blk in self.blocks for q in blk.driver_params()] + [self.norm.weight, self.norm.bias]
        y = Fn.blocks_forward_inference(tokens, self.num_heads, self.mlp_dim, prec, params, ln_eps=1e-6,
                                        qkv_bias=[blk.attn.qkv.bias for blk in self.blocks])
        return y[:, 0].contiguous()


def slice_features(volume: torch.Tensor, trunk: BiomedClipTrunk, slices: int = 32) -> torch.Tensor:
    """vit.py:529-533: volume [B,1,D,H,W] fp32 (GPU) -> trilinear resize to (32,224,224) -> each slice as a 3-channel image
    through the frozen trunk -> ``image_2d`` [B, 32, 768]."""
    Fn._need(volume)
    if volume.dim() != 5 or volume.shape[1] != 1:
        raise ValueError("slice_features expects [B, 1, D, H, W]")
    B, S = volume.shape[0], trunk.img_size
    resized = torch.stack([P.resample(volume[b, 0], (slices, S, S)) for b in range(B)])           # [B, 32, 224, 224]
    x2d = resized.view(B * slices, 1, S, S).expand(-1, 3, -1, -1).contiguous()                    # .expand(-1,3,..) of the reference
    return trunk(x2d).view(B, slices, -1)
