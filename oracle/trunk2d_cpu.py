"""TEST INFRASTRUCTURE ONLY: CPU restatement of the BiomedCLIP vision trunk = timm ``vit_base_patch16_224`` with
``num_classes=0`` (what ``model.visual.trunk`` is in multimodal_encoder/vit.py:515-518 and CT-RATE_2D_to_npy_file.py:87),
and of the slice-feature composition vit.py:529-533.

timm / open_clip are third-party and absent (no weights offline either): the architecture is restated from its published
definition (timm.models.vision_transformer: PatchEmbed Conv2d k=s=16, cls token, learned positions added to all 197
tokens, pre-LN Blocks with biased qkv and exact GELU, LayerNorm eps=1e-6, global_pool='token').  Parity unpinned against
the reference; tests/test_oracle.py pins this restatement against HuggingFace ``ViTModel`` (same architecture, independent
code) with the weights mapped across."""
import torch
import torch.nn.functional as F
from torch import nn


class Block(nn.Module):
    def __init__(self, d, heads, hidden):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(d, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(d, 3 * d, bias=True)
        self.attn.proj = nn.Linear(d, d)
        self.norm2 = nn.LayerNorm(d, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(d, hidden)
        self.mlp.fc2 = nn.Linear(hidden, d)

    def forward(self, x):
        B, N, d = x.shape
        qkv = self.attn.qkv(self.norm1(x)).reshape(B, N, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        a = ((q * (d // self.heads) ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
        x = x + self.attn.proj((a @ v).transpose(1, 2).reshape(B, N, d))
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x))))


class Trunk(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0):
        super().__init__()
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, embed_dim) * 0.02)
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(in_chans, embed_dim, patch_size, patch_size)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, int(embed_dim * mlp_ratio)) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        x = self.patch_embed.proj(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], 1) + self.pos_embed
        return self.norm(self.blocks(x))[:, 0]


def slice_features(volume, trunk, slices=32, size=224):                     # vit.py:529-533
    B = volume.shape[0]
    x = F.interpolate(volume.clone(), size=(slices, size, size), mode="trilinear", align_corners=False)
    x = x.expand(-1, 3, -1, -1, -1).permute(0, 2, 1, 3, 4)
    x = x.reshape(-1, *x.shape[-3:])
    return trunk(x).view(B, slices, -1)
