"""The reference's two CLIP train steps as fused engines.

``ClipStage1Engine`` -- M3DCLIP_stage1.forward (CLIP_stage1.py:111-139): ViT_stage1 forward + backward, cls -> mm_vision_proj ->
normalize, all-gather, InfoNCE against frozen text features, bucketed gradient all-reduce, clipped AdamW.
``ClipStage2Engine(train_stage1=True)`` -- SURVEY section 8(d) "C4-B": both encoders trainable in one iteration (6 forward units per
volume): the stage-1 step above and the stage-2 step below share ONE patch matrix, ONE stage-1 forward (its logits also feed the stage-2
relation term, detached as in the reference) and ONE feature all-gather; each model keeps its own loss (stage 1: InfoNCE with ITS
logit_scale; stage 2: InfoNCE + w * relation MSE), flat gradient, buckets and AdamW state.  The reference never trains both encoders in
one step (train_CLIP_stage1.py / train_CLIP_stage2.py): C4-B is the survey's both-trainable FLOP / all-reduce-payload variant, and each
half of it is tested against a CPU statement of its own reference step (tests/test_gpu_clip.py).

``ClipStage2Engine`` -- BASELINE configs[3] / SURVEY section 8(d) "C4-A": the reference's stage-2 CLIP train step (CLIP_stage2.py:119-151) as one
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
from .clip import M3DCLIPVision_stage1, M3DCLIPVision_stage2, relation_loss_weight
from .engine import EncoderUnit, _Engine, _al, _encoder_bounds, _encoder_entries, _engine_prec, decays
from .fused import side_stream


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


def _clip_flat(unit: EncoderUnit, model, n_buckets: int):
    """flat order of one CLIP model: encoder [head.., blocks, norm], then mm_vision_proj and logit_scale (finished FIRST by the backward
    pass); returns (entries, layer chunks, bucket bounds)."""
    dec = decays(model)
    entries = _encoder_entries(unit, "vision_encoder.")
    entries += [("mm_vision_proj.weight", model.mm_vision_proj.weight, dec["mm_vision_proj.weight"]),
                ("mm_vision_proj.bias", model.mm_vision_proj.bias, dec["mm_vision_proj.bias"]),
                ("logit_scale", model.logit_scale, dec["logit_scale"])]
    names = [n for n, _, _ in entries]
    offs, off = [], 0
    for _, p, _ in entries:
        offs.append(off)
        off += _al(p.numel())
    chunks, bounds = _encoder_bounds(unit, names, offs, "vision_encoder.", 0, off, n_buckets)
    return entries, chunks, bounds


def _clip_loss_scale(prec: int, loss_scale, n_global: int) -> float:
    if prec != L.PREC_F16:
        if loss_scale not in (None, 1, 1.0):
            raise L.HsError("loss_scale belongs to precision 'f16'")
        return 1.0
    return float(loss_scale) if loss_scale else float(2 ** (8 + max(1, int(n_global)).bit_length() - 1))


class _ClipEngine(_Engine):
    def _gather(self, local: torch.Tensor, all_feats: torch.Tensor) -> None:
        """collective C2: every rank's feature rows, in rank order.  A muted engine (``probe_step``: the collective-free fit probe) fills
        only its own rows — a rank that fails inside its probe must not leave the others waiting in a collective."""
        if self.world > 1 and not self.buckets.muted:
            dist.all_gather_into_tensor(all_feats, local, group=self.group)
        elif self.world > 1:
            all_feats.zero_()
            all_feats[self.rank * self.B:(self.rank + 1) * self.B].copy_(local)
        else:
            all_feats.copy_(local)


class ClipStage1Engine(_ClipEngine):
    """The stage-1 CLIP step (CLIP_stage1.py:111-139) over an ``M3DCLIPVision_stage1``: ``train_step(images, text_features)``;
    text features: the L2-normalised cls text features [B_local, d] of the (out-of-scope) BERT tower, frozen inputs."""

    def __init__(self, model: M3DCLIPVision_stage1, batch: int, *, lr=1e-4, weight_decay=0.1, max_grad_norm: Optional[float] = 1.0,
                 n_buckets=4, group=None, grad_wire="fp32", grad_algo="allreduce", precision: str = "bf16", loss_scale: Optional[float] = None):
        """precision 'f16': the encoder runs on IEEE fp16 operands (engine._Engine); dL/dlogits is multiplied by ``loss_scale`` (default
        2^(8 + floor(log2(global batch))): InfoNCE gradients of the cls features are O(1 / batch)) and AdamW divides it out."""
        self.prec = _engine_prec(precision)
        if model.local_loss or not model.gather_loss:
            raise L.HsError("ClipStage1Engine drives the shipped configuration: gather_loss=True, local_loss=False")
        if not all(p.requires_grad for p in model.parameters()):
            raise L.HsError("ClipStage1Engine trains every parameter of the model: requires_grad_(True) first")
        self.model, self.B, self.group = model, batch, group
        self.u = EncoderUnit(model.vision_encoder, batch, trainable=True, prec=self.prec)
        self.dev = self.u.dev
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        entries, self.chunks, bounds = _clip_flat(self.u, model, n_buckets)
        self._finish(entries, bounds, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                     grad_wire=grad_wire, grad_algo=grad_algo)
        self.u.bind(self.params)
        self.head = _Head(self.u, model.mm_vision_proj, True)
        d, n = self.u.d, batch * self.world
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.patches = torch.empty(batch * self.u.Np, self.u.Pd, dtype=self.dtype16, device=self.dev)
        self.loss_scale = _clip_loss_scale(self.prec, loss_scale, n)
        self.local = torch.empty(batch, 2 * d, **f32)              # [img | txt] of this rank
        self.all = torch.empty(n, 2 * d, **f32)                    # ... of every rank (ONE all-gather per step)
        self.raw, self.logits = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.dlog, self.draw = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.dall = torch.empty(n, d, **f32)
        self.acc = torch.zeros(2, **f32)                           # [CE loss, -]
        self.global_step = 0
        self._txt = None

    # ---- pieces shared with the both-trainable step (ClipStage2Engine(train_stage1=True)), which owns the gathered feature matrix ----
    def loss_from_all(self, all_feats: torch.Tensor, row_stride: int, col_img: int, col_txt: int) -> None:
        """logits = logit_scale * all_img @ all_txt^T from two d-wide column blocks of ``all_feats`` (CLIP_stage1.py:148-149 with
        local_loss False), symmetric InfoNCE (:151) into self.acc[0], dL/dlogits into self.dlog."""
        d, n, s = self.u.d, self.dlog.shape[0], Fn._s()
        g = L.GemmArgs()
        g.A, g.B = all_feats.data_ptr() + 4 * col_img * d, all_feats.data_ptr() + 4 * col_txt * d
        g.M, g.N, g.K, g.lda, g.ldb, g.batch = n, n, d, row_stride, row_stride, 1
        g.out_f32, g.ldc, g.alpha, g.layout, g.split_k = self.raw.data_ptr(), n, 1.0, L.GEMM_NT, 1
        L.call("hs_gemm", C.byref(g), L.PREC_F32, s)
        L.call("hs_scale_dev", self.raw.data_ptr(), self.logits.data_ptr(), self.model.logit_scale.data_ptr(), 1.0, n * n, s)
        self.acc.zero_()
        L.call("hs_clip_ce", self.logits.data_ptr(), self.dlog.data_ptr(), self.acc.data_ptr(), n, s)
        if self.loss_scale != 1.0:        # fp16 backward: every gradient of the step is linear in dL/dlogits
            L.call("hs_scale", self.dlog.data_ptr(), self.dlog.data_ptr(), self.loss_scale, n * n, s)

    def backward_from_all(self, all_feats: torch.Tensor, row_stride: int, col_txt: int) -> None:
        """self.dlog -> every gradient of this model (flat gradient zeroed here), bucketed all-reduce launched and awaited."""
        s, d, B, n = Fn._s(), self.u.d, self.B, self.dlog.shape[0]
        self._begin_backward()
        ls = self.model.logit_scale
        L.call("hs_dot", self.dlog.data_ptr(), self.raw.data_ptr(), ls.grad.data_ptr(), 1.0, n * n, s)              # d/d logit_scale
        L.call("hs_scale_dev", self.dlog.data_ptr(), self.draw.data_ptr(), ls.data_ptr(), 1.0, n * n, s)
        L.call("hs_rowmix", self.draw.data_ptr(), all_feats.data_ptr() + 4 * col_txt * d, self.dall.data_ptr(), n, n, d, n, row_stride, d, 0, s)
        if self.world > 1 and not self.buckets.muted:   # backward of the all-gather (torch.distributed.nn.all_gather): sum over ranks, keep this rank's rows
            dist.all_reduce(self.dall, op=dist.ReduceOp.SUM, group=self.group)
        L.call("hs_rows_copy", self.dall.data_ptr() + 4 * self.rank * B * d, d, self.head.df.data_ptr(), None, d, B, 1, d, s)
        self.head.backward()
        self.u.backward(self.patches, self.u.dy, self.chunks, self.buckets.reduce)
        self.buckets.wait()

    # ---- the stand-alone stage-1 step ------------------------------------------------------------------------------------------
    def forward_loss(self, images, text_features) -> None:
        s, d, B = Fn._s(), self.u.d, self.B
        Fn._need_as(text_features, "text_features", (B, d))
        self._patchify(images, self.model.vision_encoder.patch_embedding, self.patches)
        self.u.forward(self.patches)
        f = self.head.forward()
        for i, t in enumerate((f, text_features)):
            L.call("hs_rows_copy", t.data_ptr(), d, self.local.data_ptr() + 4 * i * d, None, 2 * d, B, 1, d, s)
        self._gather(self.local, self.all)
        self.loss_from_all(self.all, 2 * d, 0, 1)

    def loss_value(self) -> float:
        """host read (synchronises): loss = loss_CL_stage1."""
        return float(self.acc[0])

    def backward(self) -> None:
        self.backward_from_all(self.all, 2 * self.u.d, 1)

    def train_step(self, images, text_features=None, optimizer: bool = True) -> None:
        if text_features is None:
            if self._txt is None:
                raise L.HsError("train_step(images) needs set_text_features() first (or pass the text features)")
            text_features = self._txt
        self.forward_loss(images, text_features)
        self.backward()
        if optimizer:
            self.optimizer_step()
            self.global_step += 1

    def set_text_features(self, text_features: torch.Tensor) -> None:
        B, d = self.B, self.u.d
        if not isinstance(text_features, torch.Tensor) or tuple(text_features.shape) != (B, d):
            raise L.HsError(f"text_features must be a [B={B}, {d}] tensor")
        self._txt = text_features.detach().to(device=self.dev, dtype=torch.float32).contiguous()


class ClipStage2Engine(_ClipEngine):
    """The stage-2 CLIP step over an ``M3DCLIPVision_stage2``.  ``train_step(images, images_2d, text_features_stage1,
    text_features)``; text features: L2-normalised [B_local, d] (frozen).  ``train_stage1=True``: the both-trainable variant (module
    docstring): ``self.s1`` is a ``ClipStage1Engine`` over ``model.stage1_pretrained_CLIP`` (whose parameters must then require grad)."""

    def __init__(self, model: M3DCLIPVision_stage2, batch: int, *, lr=1e-4, weight_decay=0.1, max_grad_norm: Optional[float] = 1.0,
                 n_buckets=4, group=None, grad_wire="fp32", grad_algo="allreduce", train_stage1: bool = False, precision: str = "bf16",
                 loss_scale: Optional[float] = None):
        self.prec = _engine_prec(precision)
        if model.local_loss or not model.gather_loss:
            raise L.HsError("ClipStage2Engine drives the shipped configuration: gather_loss=True, local_loss=False")
        self.model, self.B, self.group = model, batch, group
        self.s1 = None
        if train_stage1:
            self.s1 = ClipStage1Engine(model.stage1_pretrained_CLIP, batch, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                       n_buckets=n_buckets, group=group, grad_wire=grad_wire, grad_algo=grad_algo, precision=precision,
                                       loss_scale=loss_scale)
            self.u1 = self.s1.u
        else:
            self.u1 = EncoderUnit(model.stage1_pretrained_CLIP.vision_encoder, batch, trainable=False, prec=self.prec)
        self.u2 = EncoderUnit(model.vision_encoder, batch, trainable=True, prec=self.prec)
        self.dev = self.u2.dev
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        # flat order: 2E3 encoder [head.., blocks, norm], then mm_vision_proj and logit_scale (finished FIRST by the backward pass)
        entries, self.chunks, bounds = _clip_flat(self.u2, model, n_buckets)
        self._finish(entries, bounds, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                     grad_wire=grad_wire, grad_algo=grad_algo)
        self.u2.bind(self.params)
        if self.s1 is None:
            self.u1.bind()
            self.h1 = _Head(self.u1, model.stage1_pretrained_CLIP.mm_vision_proj, False)
        else:
            self.h1 = self.s1.head
        self.h2 = _Head(self.u2, model.mm_vision_proj, True)
        u = self.u2
        d, n = u.d, batch * self.world
        f32 = dict(dtype=torch.float32, device=self.dev)
        # (both-trainable: ONE patch matrix, the stage-1 engine's -- its embedding backward reads it)
        self.patches = self.s1.patches if self.s1 is not None else torch.empty(batch * u.Np, u.Pd, dtype=self.dtype16, device=self.dev)
        self.loss_scale = _clip_loss_scale(self.prec, loss_scale, n)
        self.local = torch.empty(batch, 4 * d, **f32)              # [img1 | txt1 | img2 | txt2] of this rank
        self.all = torch.empty(n, 4 * d, **f32)                    # ... of every rank (ONE all-gather per step)
        self.raw1, self.raw2 = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.logits1, self.logits2 = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.dlog, self.draw = torch.empty(n, n, **f32), torch.empty(n, n, **f32)
        self.dall = torch.empty(n, d, **f32)
        self.acc = torch.zeros(4, **f32)                           # [CE loss, sum(dlogits*logits), relation MSE, -]
        self.global_step = 0
        self._txt1 = self._txt2 = None
        self.side = side_stream(self.dev)            # (a stream that really runs beside this one: fused._pick_concurrent_stream)

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
        self._gather(self.local, self.all)
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
        if self.loss_scale != 1.0:        # fp16 backward: scale dL/dlogits (InfoNCE + weighted relation term), AdamW takes it out
            L.call("hs_scale", self.dlog.data_ptr(), self.dlog.data_ptr(), self.loss_scale, n * n, s)
        if self.s1 is not None:                                         # the stage-1 model's own InfoNCE, with ITS logit_scale
            self.s1.loss_from_all(self.all, 4 * d, 0, 1)

    def loss_value(self) -> float:
        """host read (synchronises): loss = loss_CL_stage2 + w * loss_relation."""
        a = self.acc.tolist()
        return a[0] + self._w * a[2]

    def backward(self) -> None:
        s, d, B, n = Fn._s(), self.u2.d, self.B, self.all.shape[0]
        self._begin_backward()
        ls = self.model.logit_scale
        L.call("hs_dot", self.dlog.data_ptr(), self.raw2.data_ptr(), ls.grad.data_ptr(), 1.0, n * n, s)            # d/d logit_scale
        L.call("hs_scale_dev", self.dlog.data_ptr(), self.draw.data_ptr(), ls.data_ptr(), 1.0, n * n, s)
        L.call("hs_rowmix", self.draw.data_ptr(), self.all.data_ptr() + 4 * 3 * d, self.dall.data_ptr(), n, n, d, n, 4 * d, d, 0, s)   # d all_img = draw @ all_txt
        if self.world > 1 and not self.buckets.muted:   # backward of the all-gather (torch.distributed.nn.all_gather): sum over ranks, keep this rank's rows
            dist.all_reduce(self.dall, op=dist.ReduceOp.SUM, group=self.group)
        L.call("hs_rows_copy", self.dall.data_ptr() + 4 * self.rank * B * d, d, self.h2.df.data_ptr(), None, d, B, 1, d, s)
        self.h2.backward()
        self.u2.backward(self.patches, self.u2.dy, self.chunks, self.buckets.reduce)
        self.buckets.wait()
        if self.s1 is not None:
            self.s1.backward_from_all(self.all, 4 * d, 1)

    def optimizer_step(self) -> None:
        super().optimizer_step()
        if self.s1 is not None:
            self.s1.optimizer_step()
            self.s1.global_step += 1

    def probe_step(self, *inputs) -> None:
        if self.s1 is not None:
            self.s1.buckets.muted = True
        try:
            super().probe_step(*inputs)
        finally:
            if self.s1 is not None:
                self.s1.buckets.muted = False

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
