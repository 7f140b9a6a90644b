"""Data-parallel gradient exchange: one process per GPU, flat fp32 gradient buffer cut into contiguous
buckets, one asynchronous all-reduce(SUM) per bucket launched as soon as the backward pass has finished
writing it, so the RCCL ring/tree over xGMI overlaps the remaining backward kernels.  Device-agnostic
(``gloo`` on CPU in tests, ``nccl`` == RCCL on MI355X).  Replaces the implicit DDP hook the reference gets
from HF Trainer/accelerate (train/train_CLIP_stage1.py:88-89); mean-reduction (1/world) is folded into the
optimizer's ``grad_scale`` instead of a second pass over the gradients.

xGMI note: 8 MI355X are fully connected by 7 point-to-point links; a ring all-reduce is
bound per link, so buckets are kept large (a quarter of the encoder, ~85 MB fp32) — few, big collectives.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, flat_grad: torch.Tensor, bounds: Sequence[Tuple[int, int]], group=None, always: bool = False):
        """bounds: [(start, end)] element ranges of ``flat_grad`` in the order the backward pass completes them.
        always: launch the collectives even at world size 1 (tests: exercises the backend's asynchronous path on one GPU)."""
        self.flat = flat_grad
        self.bounds = [(int(a), int(b)) for a, b in bounds]
        for a, b in self.bounds:
            if not (0 <= a < b <= flat_grad.numel()):
                raise ValueError(f"bucket ({a},{b}) outside the flat gradient of {flat_grad.numel()} elements")
        # the buckets must PARTITION [0, total): an uncovered range would leave those gradients unsynchronised across ranks with no
        # error, an overlap would sum a range twice
        pos = 0
        for a, b in sorted(self.bounds):
            if a != pos:
                raise ValueError(f"gradient buckets {'overlap' if a < pos else 'leave a gap'} at element {min(a, pos)}: "
                                 f"they must partition [0, {flat_grad.numel()})")
            pos = b
        if pos != flat_grad.numel():
            raise ValueError(f"gradient buckets end at {pos}, the flat gradient has {flat_grad.numel()} elements")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.always = always and dist.is_available() and dist.is_initialized()
        self._pending: List = []

    @property
    def grad_scale(self) -> float:
        """multiply the summed gradient by this to get the DDP mean."""
        return 1.0 / self.world

    def reduce(self, i: int) -> None:
        """launch the all-reduce of bucket i (no-op at world size 1)."""
        if self.world == 1 and not self.always:
            return
        a, b = self.bounds[i]
        self._pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending.clear()


def gather_features(image_features: torch.Tensor, text_features: torch.Tensor, group=None):
    """utils/dist_utils.py:280-306 with gather_with_grad semantics (torch.distributed.nn.all_gather): forward
    all-gather of the [B_local, d] cls features; backward = this rank's slice of the summed gradient.  One fused
    collective for both tensors ([B_local, 2d]) instead of the reference's two.  Identity at world size 1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return image_features, text_features
    both = _AllGatherCat.apply(torch.cat([image_features, text_features], dim=1), group)
    d = image_features.shape[1]
    return both[:, :d], both[:, d:]


class _AllGatherCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        world = dist.get_world_size(group)
        out = torch.empty(world * x.shape[0], x.shape[1], dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        ctx.group, ctx.n = group, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        r = dist.get_rank(ctx.group)
        return g[r * ctx.n:(r + 1) * ctx.n], None
