"""On-disk formats around the encoder path -- host logic only, no kernels.

What the reference writes and reads for the vision towers and the Spatial Packer, restated so that checkpoints
produced by the reference scripts load into the hsenet_amd modules (and vice versa):

* CLIP pre-training checkpoints (train/train_CLIP_stage1.py:201-213, train_CLIP_stage2.py analogue): a directory
  holding the HF ``save_pretrained`` weights (``model.safetensors``) AND a plain ``torch.save`` of the full state
  dict as ``model_params.bin``.  `save_clip_checkpoint` / `load_clip_state_dict`.
* The tower hand-over into the MLLM (train/train_VLM.py:477-503): keys are filtered by substring and copied INTO
  the towers BY POSITION -- the i-th ``stage1_pretrained_CLIP`` (or, failing that, ``vision_encoder``) tensor goes to
  the i-th ``vision_tower_stage1`` tensor, the i-th remaining tensor to the i-th ``vision_tower_stage2`` tensor.
  That is why state-dict ORDER is part of the drop-in contract (138 / 150 tensors).  `filter_visual_checkpoint`,
  `load_towers_positional`.
* MLLM fine-tuning checkpoints (train/lamed_trainer.py:20-23): ``pytorch_model.bin`` with only the parameters whose
  name contains ``mm_projector`` or ``lora``; resumed with ``strict=False`` after the same two filters
  (train/train_VLM.py:505-510).  `save_projector_and_lora`, `load_projector_and_lora`.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Mapping, Tuple

import torch

SAFE_WEIGHTS_NAME = "model.safetensors"      # transformers.utils.SAFE_WEIGHTS_NAME
WEIGHTS_NAME = "pytorch_model.bin"           # transformers.utils.WEIGHTS_NAME
PARAMS_NAME = "model_params.bin"             # train_CLIP_stage1.py:211

# train_VLM.py:478 -- tensors of the CLIP wrapper that are not part of the vision towers
_NON_VISUAL = ("language_encoder", "mm_vision_proj", "mm_language_proj", "logit_scale")


def save_clip_checkpoint(out_dir: str, state_dict: Mapping[str, torch.Tensor]) -> None:
    """``saved_<step>/`` of the CLIP trainers: safetensors weights + ``model_params.bin`` (train_CLIP_stage1.py:208-211)."""
    from safetensors.torch import save_file

    os.makedirs(out_dir, exist_ok=True)
    cpu = OrderedDict((k, v.detach().to("cpu").contiguous()) for k, v in state_dict.items())
    # safetensors refuses shared storage; the towers have none, but clone defensively so views of a flat buffer save
    save_file({k: v.clone() for k, v in cpu.items()}, os.path.join(out_dir, SAFE_WEIGHTS_NAME), metadata={"format": "pt"})
    torch.save(cpu, os.path.join(out_dir, PARAMS_NAME))


def load_clip_state_dict(path: str) -> "OrderedDict[str, torch.Tensor]":
    """Read a CLIP checkpoint: a directory (prefers ``model_params.bin``, whose key order is the module order, then
    ``model.safetensors``, then ``pytorch_model.bin``) or a single file of either kind."""
    if os.path.isdir(path):
        for name in (PARAMS_NAME, SAFE_WEIGHTS_NAME, WEIGHTS_NAME):
            f = os.path.join(path, name)
            if os.path.exists(f):
                return load_clip_state_dict(f)
        raise FileNotFoundError(f"no {PARAMS_NAME} / {SAFE_WEIGHTS_NAME} / {WEIGHTS_NAME} under {path}")
    if path.endswith(".safetensors"):
        from safetensors import safe_open

        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        with safe_open(path, framework="pt", device="cpu") as f:
            for k in f.keys():          # NB: safetensors returns keys sorted by name, not in module order
                out[k] = f.get_tensor(k)
        return out
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return OrderedDict(sd)


def filter_visual_checkpoint(state_dict: Mapping[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """train_VLM.py:478: drop the text tower, the two projection heads and the logit scale."""
    return OrderedDict((k, v) for k, v in state_dict.items() if not any(s in k for s in _NON_VISUAL))


def _positional_copy(targets: Mapping[str, torch.Tensor], target_keys: List[str], ckpt: Mapping[str, torch.Tensor],
                     source_keys: List[str]) -> List[str]:
    """Copy source_keys[i] -> target_keys[i]; a missing target or a shape mismatch is skipped and reported, as the
    reference's bare ``except`` does (train_VLM.py:488-491)."""
    skipped = []
    with torch.no_grad():
        for i, k in enumerate(source_keys):
            try:
                targets[target_keys[i]].copy_(ckpt[k])
            except Exception:  # noqa: BLE001 - the reference swallows everything here
                skipped.append(k)
    return skipped


def load_towers_positional(model: torch.nn.Module, visual_ckpt: Mapping[str, torch.Tensor], vision_tower: str =
                           "vit_stage2_dual_encoders") -> Tuple[int, int, List[str]]:
    """The tower hand-over of train_VLM.py:479-503, on an already filtered checkpoint (`filter_visual_checkpoint`).

    `model` is anything whose ``state_dict()`` has keys containing ``vision_tower_stage1`` / ``vision_tower_stage2``
    (the dual tower itself, or the whole MLLM).  Returns (#stage-1 tensors, #stage-2 tensors, skipped source keys).
    A safetensors file lists its keys alphabetically: pass the ``model_params.bin`` form (module order) for this copy.
    """
    sd = model.state_dict()            # tensors alias the parameters: copy_ writes through
    skipped: List[str] = []
    n1 = n2 = 0
    if vision_tower in ("vit_single_3dvit_encoder", "vit_stage2_dual_encoders"):
        src = [k for k in visual_ckpt if "stage1_pretrained_CLIP" in k]
        if not src:                    # a stage-1 CLIP checkpoint: its only encoder IS the 3D ViT (train_VLM.py:481-484)
            src = [k for k in visual_ckpt if "vision_encoder" in k]
        dst = [k for k in sd if "vision_tower_stage1" in k]
        skipped += _positional_copy(sd, dst, visual_ckpt, src)
        n1 = len(src)
    if vision_tower in ("vit_single_2e3vit_encoder", "vit_stage2_dual_encoders"):
        src = [k for k in visual_ckpt if "stage1_pretrained_CLIP" not in k]
        dst = [k for k in sd if "vision_tower_stage2" in k]
        skipped += _positional_copy(sd, dst, visual_ckpt, src)
        n2 = len(src)
    return n1, n2, skipped


def projector_and_lora(named: Iterable[Tuple[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """lamed_trainer.py:20-22 / train_VLM.py:507-509: the tensors whose name contains ``mm_projector`` or ``lora``."""
    named = list(named)
    out = {k: v for k, v in named if "mm_projector" in k}
    out.update({k: v for k, v in named if "lora" in k})
    return out


def save_projector_and_lora(out_dir: str, model: torch.nn.Module) -> str:
    """LaMedTrainer._save: ``pytorch_model.bin`` holding only the projector (and LoRA) parameters."""
    os.makedirs(out_dir, exist_ok=True)
    f = os.path.join(out_dir, WEIGHTS_NAME)
    torch.save({k: v.detach().cpu() for k, v in projector_and_lora(model.named_parameters()).items()}, f)
    return f


def load_projector_and_lora(model: torch.nn.Module, path: str):
    """train_VLM.py:505-510 (``resume_mllm_weights``): filtered, ``strict=False``."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    return model.load_state_dict(projector_and_lora(ckpt.items()), strict=False)
