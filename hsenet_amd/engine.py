"""Train-step engines for the dual 3D vision-encoder path: forward + backward + DP gradient all-reduce + AdamW with every
kernel from libhsenet_hip.so and parameters / gradients / Adam moments in flat fp32 buffers.

* ``DualEngine``  — BASELINE configs[2]: ``ViT3DTower_dual_encoders`` (3D-ViT + 2E3-ViT on the same volume, vit.py:926-948)
  feeding two ``VisualPacker_3d_phi_v3`` whose outputs are concatenated to (B, 256, hidden) (lamed_arch.py:122-141).
* ``FlatEncoder`` — BASELINE configs[1]: one encoder (``ViT_stage1`` or ``ViT_stage2``).

These are the measurement harness' stand-in for the reference's HF Trainer step (train/train_CLIP_stage1.py:231-244 -> DDP ->
optim "adamw_torch" :85, weight_decay 0.1, default max_grad_norm 1.0): like the Trainer they exclude biases and LayerNorm
weights from weight decay and clip the global gradient norm.  The module API in encoder.py / projector.py stays the drop-in
surface; an engine ADOPTS such modules' parameters in place (they become views of the flat buffer, so ``state_dict()`` /
``load_state_dict()`` keep working and ``p.grad`` is a view of the flat gradient).

Flat-buffer order is chosen so that the backward pass completes it from the top down in contiguous ranges (packers, then per
encoder: final norm, blocks L-1..0, patch-embed/cls/pos/scorer): each finished range is all-reduced (RCCL over xGMI) while the
remaining backward kernels run.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from .dp import GradBuckets
from .encoder import ViT_stage1, ViT_stage2, ViT3DTower_dual_encoders
from .fused import chunk_ranges, optimizer_stream, side_stream
from .projector import VisualPacker_3d_phi_v3


def _al(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def bucket_plan(names, offsets, total: int, n_layers: int, n_buckets: int):
    """Backward-completion-ordered layer chunks [(l_hi, l_lo)] of ONE encoder laid out as [head.., blocks.0 .. blocks.L-1, norm]
    and the flat element range each one finishes: chunk i covers [offset(blocks.l_lo), previous start); the first chunk also
    carries the final norm (behind blocks.L-1), the last one everything in front of blocks.0 (cls / pos / patch-embed / scorer)."""
    per = max(1, (n_layers + n_buckets - 1) // n_buckets)
    chunks, hi = [], n_layers
    while hi > 0:
        lo = max(0, hi - per)
        chunks.append((hi, lo))
        hi = lo
    name_off = dict(zip(names, offsets))
    bounds, end = [], total
    for (hi, lo) in chunks:
        start = name_off[f"blocks.{lo}.mlp.linear1.weight"] if lo > 0 else 0
        bounds.append((start, end))
        end = start
    return chunks, bounds


def decays(module: nn.Module) -> dict:
    """{parameter name: bool} — HF Trainer's get_decay_parameter_names: everything except LayerNorm parameters and biases."""
    no = set()
    for mn, m in module.named_modules():
        if isinstance(m, nn.LayerNorm):
            for pn, _ in m.named_parameters(recurse=False):
                no.add(f"{mn}.{pn}" if mn else pn)
    return {n: (n not in no and "bias" not in n) for n, _ in module.named_parameters()}


class FlatParams:
    """Flat fp32 parameter / gradient / Adam-moment buffers over an ordered list of (name, parameter, decays) entries; each
    parameter starts on a 64-element boundary and becomes a view of ``flat`` (its ``.grad`` a view of ``grad``)."""

    def __init__(self, entries: Sequence[Tuple[str, nn.Parameter, bool]], device: torch.device):
        self.names = [n for n, _, _ in entries]
        offs, off = [], 0
        for _, p, _ in entries:
            offs.append(off)
            off += _al(p.numel())
        self.offsets, self.total = offs, off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        mask = torch.zeros(off // 64, dtype=torch.uint8)
        for (n, p, dec), o in zip(entries, offs):
            if p.dtype != torch.float32:
                raise L.HsError(f"parameter {n} must be fp32")
            view = self.flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
            if dec:
                mask[o // 64:(o + _al(p.numel())) // 64] = 1
        self.decay_mask = mask.to(device)
        self.gsq = torch.zeros(1, dtype=torch.float32, device=device)
        self._gsq_scratch = torch.zeros(1024, dtype=torch.float32, device=device)
        # hs_adamw_fused's device state: [0] steps applied, [1] steps skipped (non-finite gradient norm), rest scratch
        self.state = torch.zeros(8, dtype=torch.int32, device=device)
        self._param_offset = {id(p): o for (_, p, _), o in zip(entries, offs)}
        self._w16_host = torch.zeros(off // 64, dtype=torch.int64)
        self._w16_dev: Optional[torch.Tensor] = None
        self.w16_is_f16 = False
        self.check_finite = False          # fp16 engines: compute the gradient norm (and arm the skip) even without clipping
        self.grad_dirty = False            # the flat gradient holds a backward pass' values (the next backward must start from zero)

    def offset(self, name: str) -> int:
        return self.offsets[self.names.index(name)]

    def map_w16(self, param: nn.Parameter, dst_ptr: int) -> None:
        """have the optimizer pass write the 16-bit copy of `param` (a weight matrix adopted into this flat buffer) to device address
        `dst_ptr` while it updates it (hs_adamw_fused's w16_table: one destination per 64-element chunk)."""
        o, n = self._param_offset[id(param)], param.numel()
        if n % 64 or dst_ptr % 16:
            raise L.HsError("a 16-bit weight copy needs a 64-element multiple and a 16-byte aligned destination")
        self._w16_host[o // 64:(o + n) // 64] = dst_ptr + 128 * torch.arange(n // 64, dtype=torch.int64)
        self._w16_dev = None

    @property
    def step_no(self) -> int:
        """optimizer steps APPLIED (a step skipped for a non-finite fp16 gradient does not count); reads the device counter."""
        return int(self.state[0].item())

    @property
    def skipped_steps(self) -> int:
        return int(self.state[1].item())

    def zero_grad(self) -> None:
        if self.grad_dirty:
            self.grad.zero_()
            self.grad_dirty = False

    def step(self, *, lr: float, weight_decay: float, grad_scale: float, max_grad_norm: Optional[float], betas=(0.9, 0.999),
             eps: float = 1e-8, zero_grad: bool = True) -> None:
        """the optimizer pass as ONE sweep (hs_adamw_fused): decay mask, global-norm clip on the device, the gradient zeroed behind
        the read (zero_grad), the registered 16-bit weight copies written beside the fp32 masters, step / skip counters on the device."""
        if os.environ.get("HS_LEGACY_STEP") == "1":            # (A/B runs only: round 5's optimizer pass, host-side step counter)
            s = Fn._s()
            clip = max_grad_norm is not None and max_grad_norm > 0
            self._legacy_step = getattr(self, "_legacy_step", 0) + 1
            if clip:
                L.call("hs_sumsq", self.grad.data_ptr(), self.gsq.data_ptr(), self._gsq_scratch.data_ptr(), self.total, s)
            L.call("hs_adamw_masked", self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                   self.decay_mask.data_ptr(), self.total, lr, betas[0], betas[1], eps, weight_decay, grad_scale,
                   self.gsq.data_ptr() if clip else None, float(max_grad_norm) if clip else 0.0, self._legacy_step, s)
            return
        self.prepare(lr=lr, weight_decay=weight_decay, grad_scale=grad_scale, max_grad_norm=max_grad_norm, betas=betas, eps=eps,
                     zero_grad=zero_grad)
        self.sweep(0, self.total)
        if zero_grad:
            self.grad_dirty = False

    def prepare(self, *, lr: float, weight_decay: float, grad_scale: float, max_grad_norm: Optional[float], betas=(0.9, 0.999),
                eps: float = 1e-8, zero_grad: bool = True) -> None:
        """the part of the optimizer pass every sweep depends on, on the current stream: the gradient norm (when clipping, or to arm the
        fp16 skip) and hs_adamw_fused's one-thread prologue (skip decision, step counter, clip coefficient, bias corrections)."""
        s = Fn._s()
        clip = max_grad_norm is not None and max_grad_norm > 0
        gsq = None
        if clip or self.check_finite:
            L.call("hs_sumsq", self.grad.data_ptr(), self.gsq.data_ptr(), self._gsq_scratch.data_ptr(), self.total, s)
            gsq = self.gsq.data_ptr()
        if self._w16_dev is None and bool(self._w16_host.any()):
            self._w16_dev = self._w16_host.to(self.flat.device)
        self._hyper = (lr, betas[0], betas[1], eps, weight_decay, grad_scale, gsq, float(max_grad_norm) if clip else 0.0, int(zero_grad))
        L.call("hs_adamw_fused", None, None, None, None, None, 0, lr, betas[0], betas[1], eps, weight_decay, grad_scale, gsq,
               float(max_grad_norm) if clip else 0.0, int(zero_grad), None, int(self.w16_is_f16), self.state.data_ptr(), L.ADAMW_PREPARE, s)

    def sweep(self, a: int, b: int) -> None:
        """AdamW over the flat range [a, b) (64-element multiples) with the hyper-parameters of the last prepare(), on the current stream."""
        if a % 64 or b % 64 or not (0 <= a < b <= self.total):
            raise L.HsError(f"optimizer sweep range [{a}, {b}) must be 64-element aligned inside [0, {self.total})")
        lr, b1, b2, eps, wd, gs, gsq, mx, zg = self._hyper
        tab = None if self._w16_dev is None else self._w16_dev.data_ptr() + 8 * (a // 64)
        L.call("hs_adamw_fused", self.flat.data_ptr() + 4 * a, self.grad.data_ptr() + 4 * a, self.m.data_ptr() + 4 * a, self.v.data_ptr() + 4 * a,
               self.decay_mask.data_ptr() + a // 64, b - a, lr, b1, b2, eps, wd, gs, gsq, mx, zg, tab, int(self.w16_is_f16),
               self.state.data_ptr(), L.ADAMW_SWEEP, Fn._s())


def _seeds(desc, train: bool) -> None:
    """fresh Philox (seed, offset) pairs for the two dropout sites of a scorer / packer when its module is in train()."""
    if train:
        desc.drop_p = 0.1
        for i in range(2):
            sd, of = Fn.draw_dropout_seed()
            desc.seed[i], desc.offset[i] = sd, of
    else:
        desc.drop_p = 0.0


def encoder_head_params(m) -> List[Tuple[str, nn.Parameter]]:
    """embedding-driver parameters in ITS order (hs_embed_*: cls, pos, pe.w, pe.b [, scorer x12]) with their state-dict names."""
    out = []
    wb = m.patch_embedding.fused_weight_bias()
    if wb is None:
        raise L.HsError("the fused engines drive the 'perceptron' patch embedding (every shipped configuration) and the 'conv' one on "
                        "one-channel volumes; a multi-channel pos_embed='conv' encoder runs through the module API")
    conv = m.patch_embedding.pos_embed == "conv"
    if hasattr(m, "cls_token"):
        out.append(("cls_token", m.cls_token))
    out += [("patch_embedding.position_embeddings", m.patch_embedding.position_embeddings),
            ("patch_embedding.patch_embeddings.weight" if conv else "patch_embedding.patch_embeddings.1.weight", wb[0]),
            ("patch_embedding.patch_embeddings.bias" if conv else "patch_embedding.patch_embeddings.1.bias", wb[1])]
    if isinstance(m, ViT_stage2):
        out += [("patch_score_proj.weight", m.patch_score_proj.weight), ("patch_score_proj.bias", m.patch_score_proj.bias)]
        sga = m.slice_guided_attention
        for nm in ("Wq", "Wk", "Wv", "output_linear"):
            lin = getattr(sga, nm)
            out += [(f"slice_guided_attention.{nm}.weight", lin.weight), (f"slice_guided_attention.{nm}.bias", lin.bias)]
        out += [("slice_guided_attention.norm.weight", sga.norm.weight), ("slice_guided_attention.norm.bias", sga.norm.bias)]
    return out


def encoder_flat_order(m) -> List[Tuple[str, nn.Parameter]]:
    """[head.., blocks.0 .. blocks.L-1, norm]: the order the backward pass completes from the END."""
    named = dict(m.named_parameters())
    out = encoder_head_params(m)
    for l in range(len(m.blocks)):
        pre = f"blocks.{l}."
        out += [(n, p) for n, p in named.items() if n.startswith(pre)]
    out += [("norm.weight", m.norm.weight), ("norm.bias", m.norm.bias)]
    if len(out) != len(named) or {n for n, _ in out} != set(named):
        raise L.HsError("every encoder parameter must be in the flat order exactly once")
    return out


class EncoderUnit:
    """One encoder (P4 ViT_stage1 / P6 ViT_stage2) on the fused drivers: hs_embed_* + hs_vit_blocks_*, own workspaces."""

    def __init__(self, model, batch: int, trainable: bool = True, recompute_mlp: bool = False, prec: int = L.PREC_BF16):
        """recompute_mlp: do not keep the MLP's two [rows][mlp] activations per layer (46 % of the saved activation bytes: 0.6 of 1.3 GB
        per 256^3 volume and encoder); the backward re-runs fc1 + GELU per layer (hs_vit_desc.save_for_bwd = 2; gradients bit-identical).
        prec: L.PREC_BF16 (the reference's training policy) or L.PREC_F16 (fp16 operands, forward and backward; the patch matrix the
        unit is given must be in the same format)."""
        if prec not in (L.PREC_BF16, L.PREC_F16):
            raise L.HsError("the engines run on 16-bit operands: precision 'bf16' or 'f16' (the exact-f32 mode is the module API's)")
        self.prec = prec
        if not isinstance(model, (ViT_stage1, ViT_stage2)):
            raise TypeError("EncoderUnit drives ViT_stage1 / ViT_stage2")
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise L.HsError("the engine needs the module on an MI355X ('cuda') device")
        self.model, self.dev, self.B, self.trainable = model, p0.device, batch, trainable
        self.scorer = isinstance(model, ViT_stage2)
        pe = model.patch_embedding
        self.Np, self.Pd, self.d = pe.n_patches, pe.patch_dim, model.hidden_size
        self.has_cls = hasattr(model, "cls_token")
        self.N = self.Np + (1 if self.has_cls else 0)
        self.Ln = len(model.blocks)
        sv = (2 if recompute_mlp else 1) if trainable else 0
        self.desc = L.VitDesc(batch, self.N, self.d, model.num_heads, model.mlp_dim, self.Ln, prec, sv, 0.0, None)
        self._qb = self._qg = None
        if model.qkv_biases():                    # qkv_bias=True (vit.py:385): pointer tables (re-made by bind() after adoption)
            self._qb = Fn._ptr_array(model.qkv_biases())
            self.desc.qkv_bias = C.cast(self._qb, C.c_void_p)
        self.edesc = L.EmbedDesc(batch, self.Np, self.Pd, self.d, int(self.has_cls), int(self.scorer), 32, int(trainable), 0.0)
        self.edesc.prec = prec
        lib = L.load()
        # the 16-bit copies of the block weights live OUTSIDE the per-call workspace (hs_vit_desc.w16 / .w16t): made once, re-made only when
        # a parameter changes behind the engine's back (sync_weights), and kept current by the optimizer pass itself (FlatParams.map_w16:
        # hs_adamw_fused writes the straight copy as it updates; after_step() re-makes the transposed one from it)
        self.w_elems = lib.hs_vit_weights_elems(C.byref(self.desc))
        if self.w_elems == 0:
            raise L.HsError("unsupported encoder geometry")
        dt16 = torch.float16 if prec == L.PREC_F16 else torch.bfloat16
        self.w16 = torch.empty(self.w_elems, dtype=dt16, device=self.dev)
        nt = trainable and self.d % 64 == 0 and model.mlp_dim % 64 == 0
        self.w16t = torch.empty(self.w_elems, dtype=dt16, device=self.dev) if nt else None
        # HS_LEGACY_STEP=1 (A/B runs only): the round-5 step -- weight copies re-made inside the workspace by every forward / backward call,
        # gradient zeroed by a memset, hs_adamw_masked
        self.legacy = os.environ.get("HS_LEGACY_STEP") == "1"
        if self.legacy:
            self.w16 = self.w16t = None
        self.desc.w16, self.desc.w16t = Fn._p(self.w16), Fn._p(self.w16t)
        self._wmats = [w for blk in model.blocks for w in (blk.attn.qkv.weight, blk.attn.out_proj.weight, blk.mlp.linear1.weight, blk.mlp.linear2.weight)]
        self._wkey, self._flat, self._w16_mapped = None, None, False
        # forward chunks (layers [lo, hi)) that an overlapped optimizer pass releases one by one: events recorded on the optimizer stream
        # after the sweep of each chunk's parameters (engine._Engine.optimizer_step), waited for by the next forward on ITS stream
        self.fwd_chunks = chunk_ranges(self.Ln, 4)
        self._events = [torch.cuda.Event() for _ in self.fwd_chunks]
        self._pending, self._opt_done = False, None
        self.ws_bytes = lib.hs_vit_workspace_bytes(C.byref(self.desc))
        if self.ws_bytes == 0:
            raise L.HsError("unsupported encoder geometry")
        self.ws, self.ws_ptr = self._alloc(self.ws_bytes)
        f32 = dict(dtype=torch.float32, device=self.dev)
        B, N, d = batch, self.N, self.d
        self.x0 = torch.empty(B, N, d, **f32)
        self.y = torch.empty(B, N, d, **f32)
        self.hidden = torch.empty(self.Ln, B, N, d, **f32)      # the hidden-state list the reference returns (= saved layer inputs)
        if trainable:
            self.dy = torch.empty(B, N, d, **f32)
            self.dx0 = torch.empty(B, N, d, **f32)
        self._ews = None                     # sized per call: depends on drop_p (train / eval)
        self._ews_key = None

    def _alloc(self, nbytes: int):
        t = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
        return t, t.data_ptr() + (-t.data_ptr()) % 256

    # ---- parameter lists (the fused drivers' orders) ------------------------------------------------------------------
    def head_params(self) -> List[Tuple[str, nn.Parameter]]:
        return encoder_head_params(self.model)

    def flat_order(self) -> List[Tuple[str, nn.Parameter]]:
        return encoder_flat_order(self.model)

    def bind(self, params: Optional["FlatParams"] = None) -> None:
        """(re)build the device-pointer tables after the parameters were adopted into flat buffers.  params (a trainable unit): the
        flat buffers that own this encoder's weights -- their optimizer pass then writes the 16-bit copies of the block weights."""
        m = self.model
        if params is not None:
            self._flat = params.flat
            if self.trainable and not self.legacy:
                d, mlp, per = self.d, m.mlp_dim, self.w_elems // self.Ln
                base = self.w16.data_ptr()
                for l, blk in enumerate(m.blocks):
                    o = 2 * l * per
                    params.map_w16(blk.attn.qkv.weight, base + o)
                    params.map_w16(blk.attn.out_proj.weight, base + o + 2 * 3 * d * d)
                    params.map_w16(blk.mlp.linear1.weight, base + o + 2 * 4 * d * d)
                    params.map_w16(blk.mlp.linear2.weight, base + o + 2 * (4 * d * d + mlp * d))
                params.w16_is_f16 = self.prec == L.PREC_F16
                self._w16_mapped = True
        self._wkey = None
        qb = m.qkv_biases()
        if qb:
            self._qb = Fn._ptr_array(qb)
            self.desc.qkv_bias = C.cast(self._qb, C.c_void_p)
            if self.trainable:
                self._qg = Fn._ptr_array([b.grad for b in qb])
                self.desc.qkv_bias_grad = C.cast(self._qg, C.c_void_p)
        blocks = [p for blk in m.blocks for p in blk.ordered_params()]
        self._lp = Fn._ptr_array(blocks)
        self._np = Fn._ptr_array([m.norm.weight, m.norm.bias])
        head = [p for _, p in self.head_params()]
        if not self.has_cls:
            head = [None] + head
        head += [None] * (16 - len(head))
        self._ep = (C.c_void_p * 16)(*[None if p is None else p.data_ptr() for p in head])
        if self.trainable:
            self._lg = Fn._ptr_array([p.grad for p in blocks])
            self._ng = Fn._ptr_array([m.norm.weight.grad, m.norm.bias.grad])
            self._eg = (C.c_void_p * 16)(*[None if p is None else p.grad.data_ptr() for p in head])

    # ---- 16-bit weight copies --------------------------------------------------------------------------------------------------
    def _weights_key(self):
        """changes whenever torch has written one of the block weights: in place through the parameter (load_state_dict, p.mul_) or
        through the flat buffer it views; the optimizer pass writes through raw pointers and keeps the copies current itself."""
        return Fn.weights_stamp(self._wmats) + (-1 if self._flat is None else self._flat._version,)

    def flat_chunk_ranges(self, params: "FlatParams"):
        """[(a, b)] flat-buffer ranges of this encoder's parameters per forward chunk (its entries are contiguous in flat order: head..,
        blocks.0 .. blocks.L-1, norm -- encoder_flat_order): chunk 0 starts at the head, the last chunk ends behind the final norm."""
        offs = [(params._param_offset[id(p)], p.numel()) for p in self.model.parameters() if id(p) in params._param_offset]
        seg_a, seg_b = min(o for o, _ in offs), max(o + _al(n) for o, n in offs)
        starts = [seg_a] + [params._param_offset[id(self.model.blocks[lo].mlp.linear1.weight)] for lo, _ in self.fwd_chunks[1:]]
        return list(zip(starts, starts[1:] + [seg_b]))

    def sync_weights(self) -> None:
        if self.legacy:
            return
        key = self._weights_key()
        if key != self._wkey:
            if self._opt_done is not None:        # torch changed a weight while an overlapped optimizer pass may still be sweeping: wait for it
                torch.cuda.current_stream().wait_event(self._opt_done)
            L.call("hs_vit_cast_weights", C.byref(self.desc), self._lp, self.w16.data_ptr(), Fn._p(self.w16t), Fn._s())
            self._wkey = key

    def after_step(self) -> None:
        """after an optimizer pass that wrote the straight copies (map_w16): the transposed ones from them (one launch, 16-bit source)."""
        if self._w16_mapped and self.w16t is not None and self._wkey is not None:
            L.call("hs_vit_cast_weights", C.byref(self.desc), None, self.w16.data_ptr(), self.w16t.data_ptr(), Fn._s())

    # ---- passes -----------------------------------------------------------------------------------------------------------
    def _embed_ws(self):
        key = float(self.edesc.drop_p)
        if self._ews_key != key:
            nb = L.load().hs_embed_workspace_bytes(C.byref(self.edesc))
            if nb == 0:
                raise L.HsError("unsupported embedding geometry")
            self._ews, self._ews_ptr = self._alloc(nb)
            self._ews_bytes, self._ews_key = nb, key
        return self._ews_ptr, self._ews_bytes

    def forward(self, patches: torch.Tensor, image_2d: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = Fn._s()
        if self.scorer:
            if image_2d is None:
                raise L.HsError("ViT_stage2 needs image_2d (the 2D slice features)")
            if not isinstance(image_2d, torch.Tensor) or image_2d.numel() != self.B * 32 * self.d:
                raise L.HsError(f"image_2d must hold [B={self.B}, 32, hidden={self.d}] slice features")
            Fn._need(image_2d)                                          # (vit.py:332 views it as (B, 32, -1))
            Fn._need_as(image_2d.view(self.B, 32, -1), "image_2d", (self.B, 32, self.d))
            _seeds(self.edesc, self.model.training)
        self.sync_weights()
        ews, ewb = self._embed_ws()
        if self._pending:
            # an overlapped optimizer pass is (or was) updating this encoder chunk by chunk on another stream: every forward chunk waits
            # for the sweep of ITS parameters only -- the rest of the optimizer pass runs beside the first layers
            self._pending = False
            cur = torch.cuda.current_stream()
            for k, (lo, hi) in enumerate(self.fwd_chunks):
                cur.wait_event(self._events[k])
                if k == 0:
                    L.call("hs_embed_fwd", C.byref(self.edesc), patches.data_ptr(), Fn._p(image_2d) if self.scorer else None, self._ep,
                           self.x0.data_ptr(), ews, ewb, s)
                L.call("hs_vit_blocks_fwd_range", C.byref(self.desc), self.x0.data_ptr(), self._lp, self._np, self.y.data_ptr(),
                       self.hidden.data_ptr(), self.ws_ptr, self.ws_bytes, lo, hi, s)
            return self.y
        L.call("hs_embed_fwd", C.byref(self.edesc), patches.data_ptr(), Fn._p(image_2d) if self.scorer else None, self._ep,
               self.x0.data_ptr(), ews, ewb, s)
        L.call("hs_vit_blocks_fwd", C.byref(self.desc), self.x0.data_ptr(), self._lp, self._np, self.y.data_ptr(),
               self.hidden.data_ptr(), self.ws_ptr, self.ws_bytes, s)
        return self.y

    def backward(self, patches: torch.Tensor, dy: torch.Tensor, chunks: Sequence[Tuple[int, int]],
                 on_chunk: Callable[[int], None]) -> None:
        """dy [B,N,d] -> parameter gradients (accumulated into the zeroed flat gradient); ``on_chunk(i)`` after layer chunk i
        (and, for the last chunk, after the embedding backward) so the caller can launch that range's all-reduce."""
        if not self.trainable:
            raise L.HsError("this encoder unit was built frozen (forward-only)")
        Fn._need_as(dy, "dy", self.y.shape)
        s = Fn._s()
        for i, (hi, lo) in enumerate(chunks):
            L.call("hs_vit_blocks_bwd", C.byref(self.desc), self.x0.data_ptr(), self._lp, self._np, dy.data_ptr(), self.dx0.data_ptr(),
                   self._lg, self._ng, self.hidden.data_ptr(), self.ws_ptr, self.ws_bytes, hi, lo, s)
            if lo > 0:
                on_chunk(i)
        ews, ewb = self._embed_ws()
        L.call("hs_embed_bwd", C.byref(self.edesc), patches.data_ptr(), self.dx0.data_ptr(), self._ep, self._eg, ews, ewb, s)
        on_chunk(len(chunks) - 1)


class PackerUnit:
    """One Spatial Packer (P2) on hs_packer_*: reads an encoder's output in place (cls row skipped), writes its half of the
    concatenated (B, slots*cells, out_dim) tensor."""

    def __init__(self, packer: VisualPacker_3d_phi_v3, batch: int, in_tokens: int, in_row_offset: int, out_slots: int, slot: int,
                 trainable: bool = True, prec: int = L.PREC_BF16):
        p0 = next(packer.parameters())
        if not p0.is_cuda:
            raise L.HsError("the engine needs the module on an MI355X ('cuda') device")
        self.packer, self.dev, self.B, self.trainable = packer, p0.device, batch, trainable
        gd, gh, gw = packer.grid
        kd, kh, kw = packer.kernel_size
        self.cells = (gd // kd) * (gh // kh) * (gw // kw)
        self.out_dim, self.slot, self.out_slots = packer.out_dim, slot, out_slots
        d = packer.in_dim
        if in_tokens != in_row_offset + gd * gh * gw:
            raise L.HsError(f"packer grid {packer.grid} does not match {in_tokens - in_row_offset} patch tokens")
        self.desc = L.PackerDesc(batch, gd, gh, gw, kd, kh, kw, d, self.out_dim, in_row_offset, in_tokens * d,
                                 out_slots * self.cells * self.out_dim, 1 if trainable else 0, 0.0)
        self.desc.prec = prec
        self._ws_key = None

    def flat_order(self) -> List[Tuple[str, nn.Parameter]]:
        return list(self.packer.named_parameters())

    def bind(self, params: Optional["FlatParams"] = None) -> None:
        pk, ra = self.packer, self.packer.resolution_attention
        ps = [pk.proj_mpls[0].weight, pk.proj_mpls[0].bias, pk.proj_mpls[2].weight, pk.proj_mpls[2].bias,
              ra.Wq.weight, ra.Wq.bias, ra.Wk.weight, ra.Wk.bias, ra.Wv.weight, ra.Wv.bias, ra.output_linear.weight,
              ra.output_linear.bias, ra.norm.weight, ra.norm.bias]
        self._pp = Fn._ptr_array(ps)
        if self.trainable:
            self._pg = Fn._ptr_array([p.grad for p in ps])

    def _ws(self):
        key = float(self.desc.drop_p)
        if self._ws_key != key:
            nb = L.load().hs_packer_workspace_bytes(C.byref(self.desc))
            if nb == 0:
                raise L.HsError("unsupported packer geometry")
            t = torch.empty(nb + 256, dtype=torch.uint8, device=self.dev)
            self._wst, self._ws_ptr, self._ws_bytes, self._ws_key = t, t.data_ptr() + (-t.data_ptr()) % 256, nb, key
        return self._ws_ptr, self._ws_bytes

    def _slot_ptr(self, t: torch.Tensor) -> int:
        return t.data_ptr() + 4 * self.slot * self.cells * self.out_dim

    def forward(self, feat: torch.Tensor, out: torch.Tensor) -> None:
        _seeds(self.desc, self.packer.training)
        ws, wb = self._ws()
        L.call("hs_packer_fwd", C.byref(self.desc), feat.data_ptr(), self._pp, self._slot_ptr(out), ws, wb, Fn._s())

    def backward(self, dout: torch.Tensor, dfeat: Optional[torch.Tensor]) -> None:
        ws, wb = self._ws()
        L.call("hs_packer_bwd", C.byref(self.desc), self._slot_ptr(dout), self._pp, self._pg, Fn._p(dfeat), ws, wb, Fn._s())


_PRECS = {"bf16": L.PREC_BF16, "f16": L.PREC_F16}


def _engine_prec(precision: str) -> int:
    if precision not in _PRECS:
        raise L.HsError("engine precision must be 'bf16' (the reference's autocast policy) or 'f16' (fp16 operands + loss scale)")
    return _PRECS[precision]


def default_loss_scale(numel: int) -> float:
    """the power of two that makes d(mean(y^2))/dy = 2 y / numel about y itself: 2^floor(log2(numel)).  The fp16 backward pass then
    carries activation gradients of the activations' own magnitude -- well inside fp16's 6e-5 .. 65504 -- whatever the batch."""
    return float(2 ** (max(1, int(numel)).bit_length() - 1))


class _Engine:
    """shared: flat buffers, gradient buckets, optimizer.

    precision 'f16' (round 5): every 16-bit tensor of the step is IEEE fp16 instead of bf16 (same kernels, same rate, 11 instead of 8
    significant bits: forward within 1e-3 of the fp32 reference, gradients within ~2e-3).  fp16 has 5 exponent bits, so the loss
    gradient is multiplied by ``loss_scale`` (a power of two; default: default_loss_scale of the loss' element count) before the
    backward pass -- the flat gradient then holds loss_scale x the true gradient -- and the factor is taken out inside the AdamW pass
    (``hs_adamw_fused``'s grad_scale), which also skips a step whose gradient norm is not finite (what torch.cuda.amp.GradScaler
    does); ``unscaled_grad()`` returns the true gradient for inspection."""

    loss_scale = 1.0
    prec = L.PREC_BF16
    # keep_grads: leave the flat gradient (the parameters' .grad views) in place after optimizer_step() instead of zeroing it inside the
    # optimizer sweep -- what a test or a notebook that inspects gradients after a full step wants.  Default: off (HF Trainer calls
    # model.zero_grad() after every optimizer step too); HS_KEEP_GRADS=1 turns it on for every engine of the process (tests/conftest.py).
    keep_grads = False

    def _finish(self, entries, bounds, *, lr, weight_decay, max_grad_norm, group, grad_wire="fp32", grad_algo="allreduce",
                keep_grads: Optional[bool] = None):
        self.params = FlatParams(entries, self.dev)
        self.flat, self.grad = self.params.flat, self.params.grad
        self.buckets = GradBuckets(self.grad, bounds, group, wire=grad_wire, algo=grad_algo)
        self.lr, self.wd, self.max_grad_norm = lr, weight_decay, max_grad_norm
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.params.check_finite = self.prec == L.PREC_F16
        self.keep_grads = bool(keep_grads) if keep_grads is not None else os.environ.get("HS_KEEP_GRADS") == "1"

    @property
    def step_no(self) -> int:
        """optimizer steps applied (device counter: a synchronising read)."""
        return self.params.step_no

    @property
    def skipped_steps(self) -> int:
        """fp16 steps whose gradient norm was not finite: the update was skipped on the device and the step does not count
        (torch.cuda.amp.GradScaler semantics; a persistently overflowing loss_scale shows up HERE, not as silent non-training)."""
        return self.params.skipped_steps

    def _trained_units(self):
        return [u for u in self.__dict__.values() if isinstance(u, EncoderUnit) and u.trainable and u._flat is self.params.flat]

    def _begin_backward(self) -> None:
        """the flat gradient must be zero when a backward starts accumulating into it: it is after an optimizer pass (zeroed inside
        the sweep) and at construction; after a backward that was NOT followed by one (optimizer=False, keep_grads) it is cleared here."""
        self.wait_optimizer()                      # (an overlapped optimizer pass zeroes the gradient and re-makes the transposed copies)
        self.params.zero_grad()
        self.params.grad_dirty = True

    # overlap_optimizer (DualEngine / FlatEncoder; default OFF): the optimizer pass does not end the step, it opens the next one.  After
    # the norm and the one-thread prologue on the step's stream, AdamW sweeps the flat buffer range by range on a third stream in the
    # order the next forward needs the parameters (encoders' layer chunks interleaved, bottom-up; packers and everything else last) and
    # records an event per chunk; the next forward's chunks wait for their events only, the packers / the backward for the whole pass.
    # MEASURED (round 6, same process, interleaved): 14.93 -> 14.7-14.9 ms at one volume per GPU, 34.3 -> 33.9 ms at three (+0.5 / +1 %),
    # i.e. almost nothing of AdamW's 1.2 ms is hidden: the forward's GEMM blocks (8 waves x 256 VGPRs = a CU's whole register file) and
    # attention blocks (half of it) cannot be placed on a CU that holds even one AdamW wave, so the two do not really share the chip --
    # a capped grid (HS_ADAMW_SWEEP_BLOCKS) or a low-priority stream changes nothing.  Kept as an option; off by default because
    # reading parameters from another stream right after train_step() then needs wait_optimizer() (or a device synchronize).
    overlap_optimizer = False
    _opt_done = None

    def _opt_plan(self):
        """[(a, b, unit or None, chunk)] partitioning the flat buffer: the trained encoders' forward chunks round-robin, then the rest."""
        plan = getattr(self, "_opt_plan_cache", None)
        if plan is None:
            per_unit = [[(a, b, u, k) for k, (a, b) in enumerate(u.flat_chunk_ranges(self.params))] for u in self._trained_units()]
            plan = [x[k] for k in range(max((len(x) for x in per_unit), default=0)) for x in per_unit if k < len(x)]
            pos, rest = 0, []
            for a, b in sorted((a, b) for a, b, _, _ in plan):
                if a < pos:
                    raise L.HsError("optimizer plan: encoder ranges overlap")
                if a > pos:
                    rest.append((pos, a, None, -1))
                pos = b
            if pos < self.params.total:
                rest.append((pos, self.params.total, None, -1))
            plan = self._opt_plan_cache = plan + rest
        return plan

    def wait_optimizer(self) -> None:
        """order the current stream behind an overlapped optimizer pass (no-op otherwise)."""
        if self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)

    def optimizer_step(self) -> None:
        hyper = dict(lr=self.lr, weight_decay=self.wd, grad_scale=self.buckets.grad_scale / self.loss_scale, max_grad_norm=self.max_grad_norm,
                     zero_grad=not self.keep_grads)
        units = self._trained_units()
        if not self.overlap_optimizer or os.environ.get("HS_LEGACY_STEP") == "1" or not units:
            self.params.step(**hyper)
            for u in units:
                u.after_step()
            return
        self.params.prepare(**hyper)
        cur, opt = torch.cuda.current_stream(), optimizer_stream(self.dev)
        opt.wait_stream(cur)
        if self._opt_done is None:
            self._opt_done = torch.cuda.Event()
        with torch.cuda.stream(opt):
            for a, b, u, k in self._opt_plan():
                self.params.sweep(a, b)
                if u is not None:
                    u._events[k].record(opt)
            for u in units:
                u.after_step()                      # the transposed 16-bit copies (needed by the next BACKWARD)
            self._opt_done.record(opt)
        for u in units:
            u._pending, u._opt_done = True, self._opt_done
        if not self.keep_grads:
            self.params.grad_dirty = False

    def unscaled_grad(self) -> torch.Tensor:
        """the flat gradient without the loss scale (a copy when the scale is not 1)."""
        return self.grad if self.loss_scale == 1.0 else self.grad / self.loss_scale

    @property
    def dtype16(self) -> torch.dtype:
        return torch.float16 if self.prec == L.PREC_F16 else torch.bfloat16

    def probe_step(self, *inputs) -> None:
        """one forward + backward WITHOUT collectives and without the optimizer: sizes every lazily allocated workspace and surfaces an
        out-of-memory condition on THIS rank before any rank has entered a collective (bench.py's per-rank fit vote)."""
        self.buckets.muted = True
        try:
            self.train_step(*inputs, optimizer=False)
        finally:
            self.buckets.muted = False

    def _patchify(self, vol: torch.Tensor, pe, patches: torch.Tensor) -> None:
        Fn._need(vol)
        B, Cc, D, H, W = vol.shape
        if B != self.B or tuple(vol.shape[2:]) != pe.img_size or vol.dtype != torch.float32:
            raise L.HsError("volume must be contiguous fp32 on the GPU with the configured batch/geometry")
        L.call("hs_patchify", vol.data_ptr(), patches.data_ptr(), B, Cc, D, H, W, *pe.patch_size, self.prec, Fn._s())


def _encoder_entries(unit: EncoderUnit, prefix: str):
    dec = decays(unit.model)
    return [(prefix + n, p, dec[n]) for n, p in unit.flat_order()]


def _encoder_bounds(unit: EncoderUnit, params_names, params_offsets, prefix: str, seg_start: int, seg_end: int, n_buckets: int):
    per = max(1, (unit.Ln + n_buckets - 1) // n_buckets)
    chunks, hi = [], unit.Ln
    while hi > 0:
        lo = max(0, hi - per)
        chunks.append((hi, lo))
        hi = lo
    off = dict(zip(params_names, params_offsets))
    bounds, end = [], seg_end
    for (hi, lo) in chunks:
        start = off[f"{prefix}blocks.{lo}.mlp.linear1.weight"] if lo > 0 else seg_start
        bounds.append((start, end))
        end = start
    return chunks, bounds


class FlatEncoder(_Engine):
    """One encoder with flat parameter / gradient / moment buffers and an explicit fused train step (BASELINE configs[1])."""

    def __init__(self, model, batch: int, *, lr=1e-4, weight_decay=0.1, max_grad_norm: Optional[float] = 1.0, n_buckets=4, group=None,
                 grad_wire="fp32", grad_algo="allreduce", recompute_mlp: bool = False, precision: str = "bf16",
                 loss_scale: Optional[float] = None, keep_grads: Optional[bool] = None, overlap_optimizer: Optional[bool] = None):
        self.prec = _engine_prec(precision)
        self.unit = EncoderUnit(model, batch, trainable=True, recompute_mlp=recompute_mlp, prec=self.prec)
        self.model, self.dev, self.B = model, self.unit.dev, batch
        entries = _encoder_entries(self.unit, "")
        probe_names = [n for n, _, _ in entries]
        offs, off = [], 0
        for _, p, _ in entries:
            offs.append(off)
            off += _al(p.numel())
        self.layer_chunks, bounds = _encoder_bounds(self.unit, probe_names, offs, "", 0, off, n_buckets)
        self._finish(entries, bounds, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                     grad_wire=grad_wire, grad_algo=grad_algo, keep_grads=keep_grads)
        self.overlap_optimizer = bool(overlap_optimizer)
        self.unit.bind(self.params)
        u = self.unit
        self.patches = torch.empty(batch * u.Np, u.Pd, dtype=self.dtype16, device=self.dev)
        self.y, self.dy, self.hidden = u.y, u.dy, u.hidden
        if self.prec == L.PREC_F16:
            self.loss_scale = float(loss_scale) if loss_scale else default_loss_scale(self.y.numel())
        elif loss_scale not in (None, 1, 1.0):
            raise L.HsError("loss_scale belongs to precision 'f16'")

    def forward(self, vol: torch.Tensor, image_2d: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._patchify(vol, self.model.patch_embedding, self.patches)
        return self.unit.forward(self.patches, image_2d)

    def backward(self, dy: torch.Tensor) -> None:
        """dy = d(loss)/dy x self.loss_scale (1 unless precision 'f16'): the flat gradient comes out with the same factor."""
        Fn._need_as(dy, "dy", self.y.shape)
        self._begin_backward()
        self.unit.backward(self.patches, dy, self.layer_chunks, self.buckets.reduce)
        self.buckets.wait()

    def train_step(self, vol: torch.Tensor, image_2d: Optional[torch.Tensor] = None, optimizer: bool = True) -> None:
        """one step of L = mean(y^2): fwd, dL/dy = 2 y / numel, bwd (+ bucketed all-reduce), clipped AdamW (optimizer=False: without it)."""
        y = self.forward(vol, image_2d)
        L.call("hs_mse_grad_scaled", y.data_ptr(), self.dy.data_ptr(), None, y.numel(), self.loss_scale, Fn._s())
        self.backward(self.dy)
        if optimizer:
            self.optimizer_step()


class DualEngine(_Engine):
    """BASELINE configs[2]: dual encoders + two Spatial Packers -> (B, 2*cells, hidden), all four modules trainable.

    forward  = encode_images (lamed_arch.py:122-141) over ViT3DTower_dual_encoders.forward (vit.py:926-948): the volume is
               patchified ONCE, both encoders read the same bf16 patch matrix; each packer reads its encoder's output in place
               (cls row skipped = select_feature 'patch') and writes its half of the concatenated output.
    backward = packers -> 2E3 encoder -> 3D-ViT encoder, gradient ranges all-reduced as they complete."""

    def __init__(self, tower: ViT3DTower_dual_encoders, packer1: VisualPacker_3d_phi_v3, packer2: VisualPacker_3d_phi_v3, batch: int, *,
                 lr=1e-4, weight_decay=0.1, max_grad_norm: Optional[float] = 1.0, n_buckets=4, group=None,
                 two_streams: Optional[bool] = None, grad_wire="fp32", grad_algo="allreduce", recompute_mlp: bool = False,
                 precision: str = "bf16", loss_scale: Optional[float] = None, keep_grads: Optional[bool] = None,
                 overlap_optimizer: Optional[bool] = None):
        """recompute_mlp: see EncoderUnit (activation memory 3.7 -> 2.5 GB per 256^3 volume for one more GEMM per layer in the backward).
        precision / loss_scale: see _Engine."""
        self.prec = _engine_prec(precision)
        if tower.select_feature != "patch" or tower.remain_2d3d_ViT_type != "dual_vits":
            raise L.HsError("DualEngine drives the shipped configuration: select_feature='patch', remain_2d3d_ViT_type='dual_vits'")
        self.tower, self.B = tower, batch
        self.u1 = EncoderUnit(tower.vision_tower_stage1, batch, True, recompute_mlp, self.prec)
        self.u2 = EncoderUnit(tower.vision_tower_stage2, batch, True, recompute_mlp, self.prec)
        self.dev = self.u1.dev
        off1 = 1 if self.u1.has_cls else 0
        self.p1 = PackerUnit(packer1, batch, self.u1.N, off1, 2, 0, True, self.prec)
        self.p2 = PackerUnit(packer2, batch, self.u2.N, off1, 2, 1, True, self.prec)
        if packer1.out_dim != packer2.out_dim or self.p1.cells != self.p2.cells:
            raise L.HsError("the two packers must agree on out_dim and token count")
        e1 = _encoder_entries(self.u1, "vision_tower_stage1.")
        e2 = _encoder_entries(self.u2, "vision_tower_stage2.")
        d1, d2 = decays(packer1), decays(packer2)
        ep1 = [("mm_projector." + n, p, d1[n]) for n, p in self.p1.flat_order()]
        ep2 = [("mm_projector2." + n, p, d2[n]) for n, p in self.p2.flat_order()]
        entries = e1 + e2 + ep1 + ep2
        names = [n for n, _, _ in entries]
        offs, off = [], 0
        for _, p, _ in entries:
            offs.append(off)
            off += _al(p.numel())
        s2, sp, sp2 = offs[len(e1)], offs[len(e1) + len(e2)], offs[len(e1) + len(e2) + len(ep1)]
        self.chunks1, b1 = _encoder_bounds(self.u1, names, offs, "vision_tower_stage1.", 0, s2, n_buckets)
        self.chunks2, b2 = _encoder_bounds(self.u2, names, offs, "vision_tower_stage2.", s2, sp, n_buckets)
        # bucket ids: 0 = packer 2, 1 = packer 1, then the 2E3 chunks, then the 3D-ViT chunks (each in backward-completion order)
        bounds = [(sp2, off), (sp, sp2)] + b2 + b1
        self._finish(entries, bounds, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                     grad_wire=grad_wire, grad_algo=grad_algo, keep_grads=keep_grads)
        self.overlap_optimizer = bool(overlap_optimizer)
        for u in (self.u1, self.u2, self.p1, self.p2):
            u.bind(self.params)
        self.patches = torch.empty(batch * self.u1.Np, self.u1.Pd, dtype=self.dtype16, device=self.dev)
        self.out = torch.empty(batch, 2 * self.p1.cells, packer1.out_dim, dtype=torch.float32, device=self.dev)
        self.dout = torch.empty_like(self.out)
        if self.prec == L.PREC_F16:
            self.loss_scale = float(loss_scale) if loss_scale else default_loss_scale(self.out.numel())
        elif loss_scale not in (None, 1, 1.0):
            raise L.HsError("loss_scale belongs to precision 'f16'")
        # The two encoder/packer chains are independent between the shared patchify and the loss: run the 2E3 chain on a side HIP
        # stream so the dispatcher can fill CUs one chain's launch leaves idle (small batches: a 4097-row GEMM has 153-204 tiles
        # for 256 CUs, an attention launch 396 blocks for 512 slots).  Measured gain: +29 % at 1 volume per GPU, +24 % at 2, +17 % at
        # 4, +10 % at 8, +6 % at 12, +1.7 % at 20.  Default: on below 16 volumes per GPU; from 16 up the chains run on ONE stream, so
        # that per-kernel durations (HIP events, rocprofv3) are those of kernels running alone -- the roofline figures bench.py
        # reports at its default batch are not inflated by a co-running kernel.
        if two_streams is None:
            two_streams = batch < 16
        self.two_streams = two_streams
        self.side = side_stream(self.dev) if two_streams else None      # (a stream that really runs beside this one: fused._pick_concurrent_stream)

    def _fork(self):
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            return torch.cuda.stream(self.side)
        return contextlib.nullcontext()

    def _join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def forward(self, vol: torch.Tensor, image_2d: torch.Tensor) -> torch.Tensor:
        self._patchify(vol, self.tower.vision_tower_stage1.patch_embedding, self.patches)
        with self._fork():
            y2 = self.u2.forward(self.patches, image_2d)
            self.wait_optimizer()                  # (the packers' parameters are swept last by an overlapped optimizer pass)
            self.p2.forward(y2, self.out)
        y1 = self.u1.forward(self.patches)
        self.wait_optimizer()
        self.p1.forward(y1, self.out)
        self._join()
        return self.out

    def backward(self, dout: torch.Tensor) -> None:
        """dout = d(loss)/d(out) x self.loss_scale (1 unless precision 'f16'): the flat gradient comes out with the same factor."""
        Fn._need_as(dout, "dout", self.out.shape)
        self._begin_backward()
        n2 = len(self.chunks2)
        with self._fork():
            self.p2.backward(dout, self.u2.dy)
            self.buckets.reduce(0)
            self.u2.backward(self.patches, self.u2.dy, self.chunks2, lambda i: self.buckets.reduce(2 + i))
        self.p1.backward(dout, self.u1.dy)
        self.buckets.reduce(1)
        self.u1.backward(self.patches, self.u1.dy, self.chunks1, lambda i: self.buckets.reduce(2 + n2 + i))
        self._join()
        self.buckets.wait()

    def train_step(self, vol: torch.Tensor, image_2d: torch.Tensor, optimizer: bool = True) -> None:
        """one step of L = mean(out^2) over the (B, 256, hidden) projector output (optimizer=False: forward + backward + all-reduce only)."""
        out = self.forward(vol, image_2d)
        self.loss.zero_()
        L.call("hs_mse_grad_scaled", out.data_ptr(), self.dout.data_ptr(), self.loss.data_ptr(), out.numel(), self.loss_scale, Fn._s())
        self.backward(self.dout)
        if optimizer:
            self.optimizer_step()


def profile_classes() -> List[dict]:
    """read the in-library per-kernel-class timers (hs_prof_*)."""
    lib = L.load()
    out = []
    for i in range(lib.hs_prof_count()):
        cnt, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        L.call("hs_prof_read", i, C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by))
        if cnt.value:
            out.append(dict(name=lib.hs_prof_name(i).decode(), launches=cnt.value, total_ms=ms.value, flops=fl.value, bytes=by.value))
    return out
