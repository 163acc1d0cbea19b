"""Batch data-parallel helpers for the PDE layers.

The layer shards naturally: every (sample, channel) plane is independent and the parameters are
replicated, so each rank runs the fused kernels on its own slice of the batch and the only exchange
is ONE sum all-reduce of the layer's parameter gradients per step (RCCL over xGMI through
``torch.distributed``; ``backend="nccl"`` is RCCL on ROCm).  The bucket is a single flat fp32
tensor (cfg2: 1.06 MB) — latency-bound, so one collective, not one per parameter.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_range", "shard_batch", "GradBucket"]


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of ``total`` samples owned by ``rank``; the first ``total % world``
    ranks take one extra sample, so ragged batches work."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(x: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """This rank's slice of a global batch (dim 0)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    b, e = shard_range(x.shape[0], rank, world)
    return x[b:e]


class GradBucket:
    """One flat fp32 bucket for the gradients of ``params``; ``allreduce()`` sums it over the ranks
    and writes the (optionally averaged) result back into each ``.grad``.

    With a global loss that is a MEAN over the global batch and per-rank losses that are means over
    the local shard, ``average=True`` reproduces the single-process gradient when shards are equal;
    with per-rank losses that are SUMS, use ``average=False``.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)

    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def allreduce(self, average: bool = True, group=None) -> None:
        """Few launches per step: one gather of the gradients into the flat buffer, ONE collective, one scatter back."""
        views = list(self.flat.split(self.sizes))
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        for i in missing:
            self.params[i].grad = torch.zeros_like(self.params[i], dtype=torch.float32)
        grads = [p.grad for p in self.params]
        if all(g.dtype == torch.float32 and g.is_contiguous() for g in grads):
            torch.cat([g.reshape(-1) for g in grads], out=self.flat)           # one launch
        else:
            for v, g in zip(views, grads):
                v.copy_(g.reshape(-1))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))
        shaped = [v.view(p.shape) for v, p in zip(views, self.params)]
        if all(g.dtype == torch.float32 for g in grads):
            torch._foreach_copy_(grads, shaped)                               # one launch (fused foreach)
        else:
            for g, v in zip(grads, shaped):
                g.copy_(v)
