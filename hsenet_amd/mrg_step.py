"""BASELINE configs[4] / SURVEY call stack CS-3: the vision side of the MLLM finetune step —
``LamedPhi3ForCausalLM.forward`` (language_model/lamed_phi3.py:41-148) -> ``prepare_inputs_for_multimodal``
(lamed_arch.py:143-155) -> ``encode_images`` (:122-141) — as a fused engine around a caller-supplied language model.

What the reference trains in this step: the two Spatial Packers (and the LLM's LoRA adapters); the dual towers are frozen
(``requires_grad_(False)``, lamed_arch.py:66) but stay in ``train()``, so the scorer's hard-coded Dropout(0.1) sites fire.  Here:

  frozen towers forward (no activations kept) -> both packers forward (fused drivers) -> (B, 256, hidden)
  -> spliced into positions 1..256 of the token embeddings (clip.splice_image_features)
  -> ``llm_loss(inputs_embeds)``: the language model, OUT OF SCOPE (a stock HF Phi via PyTorch-ROCm, frozen)
  -> autograd gives d(inputs_embeds) -> the 256 image rows -> both packers backward -> packer gradients all-reduced -> clipped AdamW.

The LLM is a plain callable; LoRA parameters, if the caller has any, are its own business (they never enter this engine's flat buffer).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import _lib as L
from . import functional as Fn
from .clip import splice_image_features
from .encoder import ViT3DTower_dual_encoders
from .engine import EncoderUnit, PackerUnit, _Engine, _al, _engine_prec, decays
from .fused import side_stream
from .projector import VisualPacker_3d_phi_v3


class MRGStepEngine(_Engine):
    def __init__(self, tower: ViT3DTower_dual_encoders, packer1: VisualPacker_3d_phi_v3, packer2: VisualPacker_3d_phi_v3, batch: int, *,
                 lr=1e-4, weight_decay=0.0, max_grad_norm: Optional[float] = 1.0, group=None, precision: str = "bf16",
                 loss_scale: Optional[float] = None):
        """precision 'f16': towers and packers on fp16 operands (the reference's eval-time autocast dtype; engine._Engine); the caller's LLM
        loss is multiplied by ``loss_scale`` (default 2^10) before its backward and AdamW divides the packer gradients by it."""
        self.prec = _engine_prec(precision)
        if self.prec == L.PREC_F16:
            self.loss_scale = float(loss_scale) if loss_scale else 1024.0
        elif loss_scale not in (None, 1, 1.0):
            raise L.HsError("loss_scale belongs to precision 'f16'")
        if tower.select_feature != "patch" or tower.remain_2d3d_ViT_type != "dual_vits":
            raise L.HsError("MRGStepEngine drives the shipped configuration: select_feature='patch', remain_2d3d_ViT_type='dual_vits'")
        self.tower, self.B = tower, batch
        self.u1 = EncoderUnit(tower.vision_tower_stage1, batch, trainable=False, prec=self.prec)
        self.u2 = EncoderUnit(tower.vision_tower_stage2, batch, trainable=False, prec=self.prec)
        self.dev = self.u1.dev
        off1 = 1 if self.u1.has_cls else 0
        self.p1 = PackerUnit(packer1, batch, self.u1.N, off1, 2, 0, True, self.prec)
        self.p2 = PackerUnit(packer2, batch, self.u2.N, off1, 2, 1, True, self.prec)
        d1, d2 = decays(packer1), decays(packer2)
        e1 = [("mm_projector." + n, p, d1[n]) for n, p in self.p1.flat_order()]
        e2 = [("mm_projector2." + n, p, d2[n]) for n, p in self.p2.flat_order()]
        split = sum(_al(p.numel()) for _, p, _ in e1)
        total = split + sum(_al(p.numel()) for _, p, _ in e2)
        self._finish(e1 + e2, [(split, total), (0, split)], lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group)
        for u in (self.u1, self.u2, self.p1, self.p2):
            u.bind()
        self.patches = torch.empty(batch * self.u1.Np, self.u1.Pd, dtype=self.dtype16, device=self.dev)
        self.out = torch.empty(batch, 2 * self.p1.cells, packer1.out_dim, dtype=torch.float32, device=self.dev)
        self.side = side_stream(self.dev)            # (a stream that really runs beside this one: fused._pick_concurrent_stream)
        self.last_loss = None

    def encode_images(self, vol: torch.Tensor, image_2d: torch.Tensor) -> torch.Tensor:
        """lamed_arch.py:122-141, dual branch: (B, 256, hidden) — fused, no autograd graph."""
        self._patchify(vol, self.tower.vision_tower_stage1.patch_embedding, self.patches)
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.p2.forward(self.u2.forward(self.patches, image_2d), self.out)
        self.p1.forward(self.u1.forward(self.patches), self.out)
        torch.cuda.current_stream().wait_stream(self.side)
        return self.out

    def train_step(self, vol: torch.Tensor, image_2d: torch.Tensor, inputs_embeds: torch.Tensor,
                   llm_loss: Callable[[torch.Tensor], torch.Tensor]) -> None:
        """inputs_embeds: [B, S, hidden] token embeddings (``embed_tokens(input_ids)``, lamed_arch.py:151), any float dtype, no grad
        needed; llm_loss(spliced fp32 embeddings [B, S', hidden]) -> scalar loss tensor (the caller's LLM forward + LM loss)."""
        feats = self.encode_images(vol, image_2d).detach().requires_grad_()
        emb = splice_image_features(inputs_embeds.detach().float(), feats)                       # lamed_arch.py:153-154
        loss = llm_loss(emb)
        (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
        self.last_loss = loss.detach()
        if feats.grad is None:
            raise L.HsError("llm_loss(emb) must depend on the spliced image features (no gradient reached them)")
        dout = feats.grad.contiguous()
        Fn._need_as(dout, "d(image features)", self.out.shape)
        self._begin_backward()
        self.p2.backward(dout, None)                  # towers frozen: no gradient w.r.t. the encoder outputs
        self.buckets.reduce(0)
        self.p1.backward(dout, None)
        self.buckets.reduce(1)
        self.buckets.wait()
        self.optimizer_step()
