"""The encoders of the MODULE API as a short chain of fused autograd nodes (16-bit precisions).

What the reference's scripts call is ``nn.Module.forward`` -> autograd -> ``torch.optim.AdamW`` under HF Trainer / DDP
(train/train_CLIP_stage1.py:85,231-244; train_VLM.py:147-153).  This file is how that call pattern reaches the fused drivers
(hs_embed_*, hs_vit_blocks_*) without giving up what the engines get from them:

* ONE node chain per tower call, covering BOTH encoders of ``ViT3DTower_dual_encoders`` (vit.py:926-930): chunk 0's forward runs
  patch-embedding + scorer + all blocks + final norm of every encoder; the other chunks' forwards only hand out the hidden state at their
  layer boundary, so that the backward is cut into ``n_chunks`` nodes.  Parameter gradients therefore become ready chunk by chunk,
  top layers first -- DistributedDataParallel's bucketed all-reduce overlaps the rest of the backward exactly as it does for the
  reference's 12 separate TransformerBlocks.
* Inside a node the 2E3 encoder's kernels go to a side HIP stream and are joined before the node returns (below 16 volumes per GPU: a
  4097-row GEMM fills 60-80 % of the chip, an attention launch 77 %).  To autograd, DDP hooks and the optimizer every gradient is
  complete on the CALLER's stream when a node returns: no multi-stream semantics leak out of the node.
* Parameter gradients of a node are views of one zeroed flat buffer (one memset, not one per parameter); the 16-bit weight copies come
  from the module's ``W16Cache`` (re-cast when a parameter changes, not per call).

Ownership of ``blocks[l].mlp.linear2.bias``: its gradient is the column sum of the residual-stream gradient ENTERING layer l, which the
driver emits from the LayerNorm backward of layer l + 1 (or of the final norm) -- so the chunk above owns it (``_owned``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from . import functional as Fn

_SIDE = {}


def _pick_concurrent_stream(device, candidates: int = 4, beside=()) -> torch.cuda.Stream:
    """a stream whose kernels really run BESIDE the current stream's (and beside those of every stream in `beside`).  HIP multiplexes a
    process' streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default): a new stream that lands on the queue of the stream it
    is meant to overlap with runs its kernels strictly after that stream's -- the 2E3 chain then adds nothing (measured in round 6: the
    same engine at 3 volumes per GPU 87.8 or 74.3 volumes/s depending on how many streams the process had created before).  So: try a few
    consecutive pool streams, time ~1-ms spin kernels issued to the current stream, to `beside` and to the candidate, keep the first
    candidate with which they all overlap."""
    cands = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    sleep = getattr(torch.cuda, "_sleep", None)
    if sleep is None or torch.cuda.is_current_stream_capturing():      # (the probe synchronises: not inside a graph capture)
        return cands[0]
    import time
    cycles = 2_000_000                                   # ~1 ms at 2 GHz

    def timed(others) -> float:
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        sleep(cycles)
        for o in others:                                 # (no dependencies: the spins must be free to overlap)
            with torch.cuda.stream(o):
                sleep(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    with torch.cuda.device(device):
        timed(())                                        # warm-up (first launch, lazy init)
        one = min(timed(()) for _ in range(2))
        best, best_t = cands[0], None
        for c in cands:
            t = min(timed(tuple(beside) + (c,)) for _ in range(2))
            if best_t is None or t < best_t:
                best, best_t = c, t
            if t < 1.5 * one:                            # all spins overlapped
                return c
    return best


def _key(device):
    device = torch.device(device) if not isinstance(device, torch.device) else device
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device, (device.index, torch.cuda.current_stream(device).cuda_stream)


def side_stream(device) -> torch.cuda.Stream:
    """the process-wide side stream of `device` for the stream that is current now (engines and the module API share it)."""
    device, key = _key(device)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = _pick_concurrent_stream(device)
    return s


def optimizer_stream(device) -> torch.cuda.Stream:
    """a third stream that runs beside BOTH the current stream and side_stream(device): the engines' optimizer sweeps overlap the next
    forward's two chains on it."""
    device, key = _key(device)
    key = key + ("opt",)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = _pick_concurrent_stream(device, beside=(side_stream(device),))
    return s


def chunk_ranges(n_layers: int, n_chunks: int):
    """[(lo, hi)] bottom-up: chunk k covers layers [lo, hi); the engines' split (engine.bucket_plan) seen from the forward side."""
    per = max(1, (n_layers + n_chunks - 1) // n_chunks)
    out, hi = [], n_layers
    while hi > 0:
        lo = max(0, hi - per)
        out.append((lo, hi))
        hi = lo
    return out[::-1]


class EncJob:
    """one encoder's pass through the fused drivers for one forward call: descriptors, parameter tables, workspaces, outputs."""

    def __init__(self, module, x: torch.Tensor, image_2d: Optional[torch.Tensor], prec: int):
        pe = module.patch_embedding
        if tuple(x.shape[2:]) != pe.img_size:
            raise ValueError(f"input volume {tuple(x.shape[2:])} does not match img_size {pe.img_size}")
        self.module, self.prec = module, prec
        self.scorer = image_2d is not None
        self.B, self.Np, self.Pd, self.d = x.shape[0], pe.n_patches, pe.patch_dim, module.hidden_size
        self.head = [None if h is None else h.contiguous() for h in module._head16()]
        self.has_cls = self.head[0] is not None
        self.N, self.Ln = self.Np + int(self.has_cls), len(module.blocks)
        self.blocks = [p.contiguous() for blk in module.blocks for p in blk.ordered_params()]
        self.norm = [module.norm.weight.contiguous(), module.norm.bias.contiguous()]
        self.qbias = [b.contiguous() for b in module.qkv_biases()]       # qkv_bias=True encoders (vit.py:385): one [3d] bias per layer
        every = [h for h in self.head if h is not None] + self.blocks + self.norm + self.qbias
        Fn._need(*every)
        self.need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in every)
        self.patches = Fn.shared_patches.get(x, pe.patch_size, prec)
        self.sem = None
        if self.scorer:
            self.sem = Fn._f32(image_2d.to(x.device)).reshape(self.B, 32, -1).contiguous()       # vit.py:332
            Fn._need_as(self.sem, "image_2d", (self.B, 32, self.d))
        dev = self.patches.device
        sv = int(self.need_grad)
        self.edesc = L.EmbedDesc(self.B, self.Np, self.Pd, self.d, int(self.has_cls), int(self.scorer), 32, sv, 0.0)
        self.edesc.prec = prec
        if self.scorer:
            Fn._set_dropout(self.edesc, module.training)
        self._qb = Fn._ptr_array(self.qbias) if self.qbias else None
        self.desc = L.VitDesc(self.B, self.N, self.d, module.num_heads, module.mlp_dim, self.Ln, prec, sv, 0.0,
                              C.cast(self._qb, C.c_void_p) if self.qbias else None)
        self.lp, self.np_ = Fn._ptr_array(self.blocks), Fn._ptr_array(self.norm)
        self.ep = Fn._opt_ptr_array(self.head, 16)
        cache = module.__dict__.get("_w16cache")
        if cache is None:
            cache = module.__dict__["_w16cache"] = Fn.W16Cache()
        mats = [self.blocks[l * 11 + i] for l in range(self.Ln) for i in (8, 6, 0, 2)]            # qkv, out_proj, linear1, linear2
        self.w16, self.w16t = cache.get(self.desc, mats, self.lp, prec, self.need_grad)
        self.desc.w16, self.desc.w16t = self.w16.data_ptr(), Fn._p(self.w16t)
        lib = L.load()
        self.ews_bytes = lib.hs_embed_workspace_bytes(C.byref(self.edesc))
        self.ws_bytes = lib.hs_vit_workspace_bytes(C.byref(self.desc))
        if self.ews_bytes == 0 or self.ws_bytes == 0:
            raise L.HsError(f"unsupported encoder geometry B={self.B} N={self.N} d={self.d} L={self.Ln}")
        self.ews, self.ews_off = Fn._aligned_bytes(self.ews_bytes, dev)
        self.ws, self.ws_off = Fn._aligned_bytes(self.ws_bytes, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.x0 = torch.empty(self.B, self.N, self.d, **f32)
        self.y = torch.empty(self.B, self.N, self.d, **f32)
        self.hidden = torch.empty(self.Ln, self.B, self.N, self.d, **f32)
        self.dx0 = None

    # ---- the two drivers, on whatever stream is current ------------------------------------------------------------------------------
    def forward(self) -> None:
        s = Fn._s()
        L.call("hs_embed_fwd", C.byref(self.edesc), self.patches.data_ptr(), Fn._p(self.sem), self.ep, self.x0.data_ptr(),
               self.ews.data_ptr() + self.ews_off, self.ews_bytes, s)
        L.call("hs_vit_blocks_fwd", C.byref(self.desc), self.x0.data_ptr(), self.lp, self.np_, self.y.data_ptr(), self.hidden.data_ptr(),
               self.ws.data_ptr() + self.ws_off, self.ws_bytes, s)

    def owned(self, lo: int, hi: int) -> List[Optional[torch.Tensor]]:
        """the parameters whose gradients the backward of layers [lo, hi) COMPLETES, in owned_slots' order (None = an embedding-driver
        slot this module does not have)."""
        src = {"b": lambda l, i: self.blocks[l * 11 + i], "n": lambda _, i: self.norm[i], "h": lambda _, i: self.head[i],
               "q": lambda l, _: self.qbias[l]}
        return [src[kind](l, i) for kind, l, i in owned_slots(self.Ln, lo, hi, bool(self.qbias))]

    def backward(self, lo: int, hi: int, dy: Optional[torch.Tensor], grads: Sequence[Optional[torch.Tensor]]) -> None:
        """layers [lo, hi) (+ the final norm for hi == L, the embedding for lo == 0); grads: zeroed tensors in owned(lo, hi)'s order."""
        lg, ng, eg = (C.c_void_p * (self.Ln * 11))(), (C.c_void_p * 2)(), (C.c_void_p * 16)()
        qg = (C.c_void_p * self.Ln)()
        for (kind, l, i), g in zip(owned_slots(self.Ln, lo, hi, bool(self.qbias)), grads):
            if g is None:
                continue
            if kind == "b":
                lg[l * 11 + i] = g.data_ptr()
            elif kind == "n":
                ng[i] = g.data_ptr()
            elif kind == "q":
                qg[l] = g.data_ptr()
            else:
                eg[i] = g.data_ptr()
        self.desc.qkv_bias_grad = C.cast(qg, C.c_void_p) if self.qbias else None
        s = Fn._s()
        L.call("hs_vit_blocks_bwd", C.byref(self.desc), self.x0.data_ptr(), self.lp, self.np_, Fn._p(dy), Fn._p(self.dx0), lg, ng,
               self.hidden.data_ptr(), self.ws.data_ptr() + self.ws_off, self.ws_bytes, hi, lo, s)
        if lo == 0:
            L.call("hs_embed_bwd", C.byref(self.edesc), self.patches.data_ptr(), self.dx0.data_ptr(), self.ep, eg,
                   self.ews.data_ptr() + self.ews_off, self.ews_bytes, s)


def owned_slots(n_layers: int, lo: int, hi: int, qkv_bias: bool = False):
    """[(kind, layer, index)] of the parameter gradients that the backward of layers [lo, hi) completes, in a fixed order:
    ('b', l, i) block parameter i (MONAI order, TransformerBlock.ordered_params) of layer l; ('n', 0, i) the final norm's weight / bias;
    ('h', 0, i) slot i of the embedding driver's 16; ('q', l, 0) layer l's qkv bias (qkv_bias=True encoders).  A layer's linear2.bias (i = 3) belongs to the chunk ABOVE it: its gradient is the
    column sum of the residual-stream gradient entering the layer, which the driver emits from the LayerNorm backward of layer l + 1 (the
    final norm's for the top layer)."""
    out = []
    for l in range(lo, hi):
        for i in range(11):
            if i == 3 and l == hi - 1 and hi < n_layers:
                continue
            out.append(("b", l, i))
        if qkv_bias:
            out.append(("q", l, 0))
    if hi == n_layers:
        out += [("n", 0, 0), ("n", 0, 1)]
    if lo > 0:
        out.append(("b", lo - 1, 3))
    else:
        out += [("h", 0, i) for i in range(16)]
    return out


def _fork(jobs: Sequence[EncJob], fn, two_streams: bool) -> None:
    """fn(job) for every job; with two_streams the second job's kernels go to the side stream and are joined before returning."""
    if not two_streams or len(jobs) < 2:
        for j in jobs:
            fn(j)
        return
    cur = torch.cuda.current_stream()
    side = side_stream(jobs[0].patches.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for j in jobs[1:]:
            fn(j)
    fn(jobs[0])
    cur.wait_stream(side)


class _Run:
    """state shared by the chunk nodes of one tower call."""

    def __init__(self, jobs, ranges, two_streams):
        self.jobs, self.ranges, self.two_streams = jobs, ranges, two_streams


class ChunkFn(torch.autograd.Function):
    """chunk k of the chain.  inputs: one tensor per job that ties the chunk to the one below (chunk 0: the parameters only), then the
    parameters it owns, job by job.  outputs: per job the hidden state at the chunk's upper boundary (the post-norm output for the top
    chunk)."""

    @staticmethod
    def forward(ctx, run: _Run, k: int, *tensors):
        jobs = run.jobs
        lo, hi = run.ranges[k]
        if k == 0:
            _fork(jobs, EncJob.forward, run.two_streams)
        ctx.run, ctx.k = run, k
        nj = len(jobs)
        ctx.save_for_backward(*[t for t in tensors[nj:] if t is not None])     # (versions checked by torch: no backward on edited weights)
        top = k == len(run.ranges) - 1
        # (fresh aliases: handing out j.y itself would give it this node as grad_fn -- ctx -> run -> job -> y -> node -> ctx, a cycle
        # through C++ that Python's collector cannot break: every step's activations would leak)
        outs = tuple(j.y.view(j.y.shape) if top else j.hidden[hi - 1] for j in jobs)
        return outs[0] if nj == 1 else outs

    @staticmethod
    def backward(ctx, *dys):
        run, k = ctx.run, ctx.k
        _ = ctx.saved_tensors                                           # raises if a saved parameter was modified in place
        jobs = run.jobs
        lo, hi = run.ranges[k]
        top = k == len(run.ranges) - 1
        live = [j for j in jobs if j.need_grad]
        owned = {id(j): j.owned(lo, hi) for j in live}
        flat = Fn._flat_grads([t for j in live for t in owned[id(j)]])
        grads, pos = {}, 0
        for j in live:
            n = len(owned[id(j)])
            grads[id(j)] = flat[pos:pos + n]
            pos += n
        dyc = {}
        for j, dy in zip(jobs, dys):
            if top and j.need_grad:
                dy = Fn._f32(dy).contiguous()
                Fn._need(dy)
                dyc[id(j)] = dy
            if lo == 0 and j.need_grad and j.dx0 is None:
                j.dx0 = torch.empty_like(j.x0)

        _fork(live, lambda j: j.backward(lo, hi, dyc.get(id(j)), grads[id(j)]), run.two_streams)
        # gradients w.r.t. the chain inputs: the real state travels inside the drivers' workspaces (hs_vit_blocks_bwd continues from the
        # residual-stream gradient the chunk above left there), so the edges carry placeholders of the right shape
        ins = []
        if k > 0:
            z = torch.zeros((), dtype=torch.float32, device=jobs[0].y.device)
            ins = [z.expand(j.hidden[lo - 1].shape) for j in jobs]
        else:
            ins = [None] * len(jobs)
        out = []
        for j in jobs:
            out += list(grads[id(j)]) if j.need_grad else [None] * len(j.owned(lo, hi))
        return (None, None, *ins, *out)


def run_encoders(jobs: Sequence[EncJob], n_chunks: int = 4, two_streams: bool = False):
    """-> [(y, [hidden states])] per job.  Without any trainable parameter (eval / frozen towers): the plain forward, no nodes."""
    if len({(j.Ln, j.B) for j in jobs}) != 1:
        raise L.HsError("encoders run together must agree on depth and batch")
    two_streams = bool(two_streams) and len(jobs) > 1
    if not any(j.need_grad for j in jobs):
        _fork(jobs, EncJob.forward, two_streams)
        return [(j.y, [j.hidden[i] for i in range(j.Ln)]) for j in jobs]
    for j in jobs:
        j.module = None          # (nothing below needs it; a job lives as long as its graph)
    ranges = chunk_ranges(jobs[0].Ln, n_chunks)
    run = _Run(list(jobs), ranges, two_streams)
    dummy = torch.zeros((), dtype=torch.float32, device=jobs[0].y.device)
    carry = [dummy] * len(jobs)
    for k, (lo, hi) in enumerate(ranges):
        params = [t for j in jobs for t in j.owned(lo, hi)]
        res = ChunkFn.apply(run, k, *carry, *params)
        carry = [res] if len(jobs) == 1 else list(res)
    return [(y, [j.hidden[i] for i in range(j.Ln)]) for j, y in zip(jobs, carry)]


def dual_streams_default(batch: int) -> bool:
    """the engines' rule (engine.DualEngine): two streams below 16 volumes per GPU; HS_DUAL_STREAMS=0 / 1 overrides."""
    env = os.environ.get("HS_DUAL_STREAMS")
    return batch < 16 if env is None else env != "0"
