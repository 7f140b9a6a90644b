"""hsenet_amd — MI355X (gfx950) implementation of HSENet's dual 3D vision-encoder path.

Public surface = the reference's own LaMed encoder/projector API (see encoder.py / projector.py) backed by the
C ABI in include/hsenet_hip.h.  Importing the package never touches the GPU; calling a module without the built
HIP library or with CPU tensors raises (there is no fallback path).
"""
from .functional import set_precision, get_precision, invalidate_weight_copies  # noqa: F401
from .encoder import (ViT_stage1, ViT_stage2, ViT3DTower_dual_encoders, regular_attention,  # noqa: F401
                      build_vision_tower, PatchEmbeddingBlock, TransformerBlock)
from .projector import (VisualPacker_3d_phi_v3, resolution_attention_v3, build_mm_projector, encode_images, LamedMetaForCausalLM)  # noqa: F401
from .clip import encode_image_cls, image_text_contrastive_learning  # noqa: F401

__all__ = ["ViT_stage1", "ViT_stage2", "ViT3DTower_dual_encoders", "regular_attention", "build_vision_tower",
           "VisualPacker_3d_phi_v3", "resolution_attention_v3", "build_mm_projector", "encode_images", "LamedMetaForCausalLM",
           "encode_image_cls", "image_text_contrastive_learning", "set_precision", "get_precision", "invalidate_weight_copies"]
from .trunk2d import BiomedClipTrunk, slice_features  # noqa: F401,E402
