"""BASELINE configs[3] / SURVEY section 8(d) "C4-A": the reference's stage-2 CLIP train step (CLIP_stage2.py:119-151) as one
fused engine — frozen stage-1 3D-ViT forward (no activations kept: the reference runs it under inference_mode, :124-127),
trainable 2E3-ViT forward + backward, cls -> mm_vision_proj -> normalize, all-gather of the cls features across ranks (collective
C2), InfoNCE + relation MSE with the 0.1*(1 - step/5000) schedule, bucketed all-reduce of the 2E3 encoder + mm_vision_proj +
logit_scale gradients overlapped with the backward pass (collective C1), clipped AdamW.  The BERT text towers are out of scope
(frozen text embeddings are inputs).  Every kernel comes from libhsenet_hip.so; the logit scale is read on the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib as L
from . import functional as Fn
from .clip import M3DCLIPVision_stage2, relation_loss_weight
from .engine import EncoderUnit, _Engine, _al, _encoder_bounds, _encoder_entries, decays


def make_clip_stage2(img_size, patch_size, **kw) -> M3DCLIPVision_stage2:
    return M3DCLIPVision_stage2(img_size, patch_size, **kw)


class _Head:
    """cls -> mm_vision_proj -> F.normalize for one encoder unit (CLIP_stage1.py:97-102,117), exact-f32 MFMA."""

    def __init__(self, unit: EncoderUnit, proj: torch.nn.Linear, trainable: bool):
        self.u, self.proj, self.trainable = unit, proj, trainable
        B, d, dev = unit.B, unit.d, unit.dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.cls = torch.empty(B, d, **f32)
        self.z = torch.empty(B, d, **f32)
        self.f = torch.empty(B, d, **f32)
        if trainable:
            self.df = torch.empty(B, d, **f32)
            self.dz = torch.empty(B, d, **f32)
            self.dcls = torch.empty(B, d, **f32)

    def forward(self) -> torch.Tensor:
        u, s = self.u, Fn._s()
        L.call("hs_rows_copy", u.y.data_ptr(), u.N * u.d, self.cls.data_ptr(), None, u.d, u.B, 1, u.d, s)           # [:, 0]
        Fn.gemm(self.cls, self.proj.weight, L.GEMM_NT, prec=L.PREC_F32, out_f32=self.z, bias=self.proj.bias, flags=L.EPI_BIAS)
        L.call("hs_l2norm_fwd", self.z.data_ptr(), self.f.data_ptr(), u.B, u.d, 1e-12, s)
        return self.f

    def backward(self) -> None:
        """self.df (gradient of the normalised feature) -> proj gradients (accumulated) and u.dy (only the cls row is non-zero)."""
        u, s = self.u, Fn._s()
        L.call("hs_l2norm_bwd", self.z.data_ptr(), self.df.data_ptr(), self.dz.data_ptr(), u.B, u.d, 1e-12, s)
        Fn.gemm(self.dz, self.cls, L.GEMM_TN, prec=L.PREC_F32, out_f32=self.proj.weight.grad, flags=L.EPI_ATOMIC)
        Fn.colsum(self.dz, self.proj.bias.grad)
        Fn.gemm(self.dz, self.proj.weight, L.GEMM_NN, prec=L.PREC_F32, out_f32=self.dcls)
        L.call("hs_rows_scatter", self.dcls.data_ptr(), u.dy.data_ptr(), u.B, 0, 1, u.N, u.d, s)


class ClipStage2Engine(_Engine):
    """The stage-2 CLIP step over an ``M3DCLIPVision_stage2``.  ``train_step(images, images_2d, text_features_stage1,
    text_features)``; text features: L2-normalised [B_local, d] (frozen)."""

    def __init__(self, model: M3DCLIPVision_stage2, batch: int, *, lr=1e-4, weight_decay=0.1, max_grad_norm: Optional[float] = 1.0,
                 n_buckets=4, group=None, grad_wire="fp32", grad_algo="allreduce"):
        if model.local_loss or not model.gather_loss:
            raise L.HsError("ClipStage2Engine drives the shipped configuration: gather_loss=True, local_loss=False")
        self.model, self.B, self.group = model, batch, group
        self.u1 = EncoderUnit(model.stage1_pretrained_CLIP.vision_encoder, batch, trainable=False)
        self.u2 = EncoderUnit(model.vision_encoder, batch, trainable=True)
        self.dev = self.u2.dev
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        # flat order: 2E3 encoder [head.., blocks, norm], then mm_vision_proj and logit_scale (finished FIRST by the backward pass)
        dec = decays(model)
        entries = _encoder_entries(self.u2, "vision_encoder.")
        entries += [("mm_vision_proj.weight", model.mm_vision_proj.weight, dec["mm_vision_proj.weight"]),
                    ("mm_vision_proj.bias", model.mm_vision_proj.bias, dec["mm_vision_proj.bias"]),
                    ("logit_scale", model.logit_scale, dec["logit_scale"])]
        names = [n for n, _, _ in entries]
        offs, off = [], 0
        for _, p, _ in entries:
            offs.append(off)
            off += _al(p.numel())
        self.chunks, bounds = _encoder_bounds(self.u2, names, offs, "vision_encoder.", 0, off, n_buckets)
        self._finish(entries, bounds, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                     grad_wire=grad_wire, grad_algo=grad_algo)
        self.u1.bind()
        self.u2.bind()
        self.h1 = _Head(self.u1, model.stage1_pretrained_CLIP.mm_vision_proj, False)
        self.h2 = _Head(self.u2, model.mm_vision_proj, True)
        u = self.u2
        d, n = u.d, batch * self.world
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.patches = torch.empty(batch * u.Np, u.Pd, dtype=torch.bfloat16, device=self.dev)
        self.local = torch.empty(batch, 4 * d, **f32)              # [img1 | txt1 | img2 | txt2] of this rank
        self.all = torch.empty(n, 4 * d, **f32)                    # ... of every rank (ONE all-gather per step)
        self.raw1, self.raw2 = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.logits1, self.logits2 = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.dlog, self.draw = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.dall = torch.empty(n, d, **f32)
        self.acc = torch.zeros(4, **f32)                           # [CE loss, sum(dlogits*logits), relation MSE, -]
        self.global_step = 0
        self._txt1 = self._txt2 = None
        self.side = torch.cuda.Stream(device=self.dev)

    # ------------------------------------------------------------------------------------------------------------------------
    def _logits(self, col_img: int, col_txt: int, raw: torch.Tensor, logits: torch.Tensor) -> None:
        """logits = logit_scale * all_img @ all_txt^T from column blocks of self.all (row stride 4d), scale read on the device."""
        d, n = self.u2.d, self.all.shape[0]
        g = L.GemmArgs()
        g.A, g.B = self.all.data_ptr() + 4 * col_img * d, self.all.data_ptr() + 4 * col_txt * d
        g.M, g.N, g.K, g.lda, g.ldb, g.batch = n, n, d, 4 * d, 4 * d, 1
        g.out_f32, g.ldc, g.alpha, g.layout, g.split_k = raw.data_ptr(), n, 1.0, L.GEMM_NT, 1
        L.call("hs_gemm", C.byref(g), L.PREC_F32, Fn._s())
        L.call("hs_scale_dev", raw.data_ptr(), logits.data_ptr(), self.model.logit_scale.data_ptr(), 1.0, n * n, Fn._s())

    def forward_loss(self, images, images_2d, text_features_stage1, text_features) -> None:
        """everything up to dL/dlogits (kept in self.dlog); loss pieces accumulate in self.acc."""
        s, d, B = Fn._s(), self.u2.d, self.B
        for nm, t in (("text_features_stage1", text_features_stage1), ("text_features", text_features)):
            Fn._need_as(t, nm, (B, d))              # hs_rows_copy reads B*d fp32 values through the raw pointer
        self._patchify(images, self.model.vision_encoder.patch_embedding, self.patches)
        # the frozen tower on a side stream: independent of the 2E3 forward until the logits
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.u1.forward(self.patches)                               # frozen stage-1 CLIP image tower (:124-127)
            f1 = self.h1.forward()
        self.u2.forward(self.patches, images_2d)                        # trainable 2E3 encoder (:131)
        f2 = self.h2.forward()
        torch.cuda.current_stream().wait_stream(self.side)
        for i, t in enumerate((f1, text_features_stage1, f2, text_features)):
            L.call("hs_rows_copy", t.data_ptr(), d, self.local.data_ptr() + 4 * i * d, None, 4 * d, B, 1, d, s)
        if self.world > 1:
            dist.all_gather_into_tensor(self.all, self.local, group=self.group)                          # collective C2
        else:
            self.all.copy_(self.local)
        self._logits(0, 1, self.raw1, self.logits1)                     # stage-1 logits with THIS model's logit_scale (:128), detached
        self._logits(2, 3, self.raw2, self.logits2)
        n = self.all.shape[0]
        self.acc.zero_()
        L.call("hs_clip_ce", self.logits2.data_ptr(), self.dlog.data_ptr(), self.acc.data_ptr(), n, s)    # InfoNCE (:137)
        w = relation_loss_weight(self.global_step)
        # relation MSE (:139,187-190): logits_per_text = logits_per_image.T, so the two mse terms are equal
        L.call("hs_relation_mse", self.logits1.data_ptr(), self.logits2.data_ptr(), self.dlog.data_ptr(), self.acc.data_ptr() + 8, n * n,
               float(w), 1, s)
        self._w = w

    def loss_value(self) -> float:
        """host read (synchronises): loss = loss_CL_stage2 + w * loss_relation."""
        a = self.acc.tolist()
        return a[0] + self._w * a[2]

    def backward(self) -> None:
        s, d, B, n = Fn._s(), self.u2.d, self.B, self.all.shape[0]
        self.grad.zero_()
        ls = self.model.logit_scale
        L.call("hs_dot", self.dlog.data_ptr(), self.raw2.data_ptr(), ls.grad.data_ptr(), 1.0, n * n, s)            # d/d logit_scale
        L.call("hs_scale_dev", self.dlog.data_ptr(), self.draw.data_ptr(), ls.data_ptr(), 1.0, n * n, s)
        L.call("hs_rowmix", self.draw.data_ptr(), self.all.data_ptr() + 4 * 3 * d, self.dall.data_ptr(), n, n, d, n, 4 * d, d, 0, s)   # d all_img = draw @ all_txt
        if self.world > 1:       # backward of the all-gather (torch.distributed.nn.all_gather): sum over ranks, keep this rank's rows
            dist.all_reduce(self.dall, op=dist.ReduceOp.SUM, group=self.group)
        L.call("hs_rows_copy", self.dall.data_ptr() + 4 * self.rank * B * d, d, self.h2.df.data_ptr(), None, d, B, 1, d, s)
        self.h2.backward()
        self.u2.backward(self.patches, self.u2.dy, self.chunks, self.buckets.reduce)
        self.buckets.wait()

    def train_step(self, images, images_2d, text_features_stage1=None, text_features=None, optimizer: bool = True) -> None:
        if (text_features_stage1 is None) != (text_features is None):
            raise L.HsError("train_step: pass BOTH text feature tensors or neither (the cached pair of set_text_features would silently "
                            "replace the one that was given)")
        if text_features_stage1 is None:
            if self._txt1 is None:
                raise L.HsError("train_step(images, images_2d) needs set_text_features() first (or pass both text feature tensors)")
            text_features_stage1, text_features = self._txt1, self._txt2
        self.forward_loss(images, images_2d, text_features_stage1, text_features)
        self.backward()
        if optimizer:
            self.optimizer_step()
            self.global_step += 1

    def set_text_features(self, text_features_stage1: torch.Tensor, text_features: torch.Tensor) -> None:
        """frozen text embeddings reused by every ``train_step(images, images_2d)`` call (bench.py)."""
        B, d = self.B, self.u2.d
        txt = []
        for nm, t in (("text_features_stage1", text_features_stage1), ("text_features", text_features)):
            if not isinstance(t, torch.Tensor) or tuple(t.shape) != (B, d):
                raise L.HsError(f"{nm} must be a [B={B}, {d}] tensor")
            t = t.detach().to(device=self.dev, dtype=torch.float32).contiguous()      # an autocast text tower hands over bf16 / fp16
            Fn._need_as(t, nm, (B, d))
            txt.append(t)
        self._txt1, self._txt2 = txt
