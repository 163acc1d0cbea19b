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
    with per-rank losses that are SUMS (ragged shards), use ``average=False``.

    Launch count per step, from most to fewest:
      * plain ``allreduce()``: one gather (``cat``) into the flat buffer, the collective, one fused copy back
        (the division of ``average`` rides on the collective where the backend has ``ReduceOp.AVG`` — RCCL does);
      * ``grads_as_views=True`` (or ``install_views()``): every ``.grad`` IS a slice of the flat buffer, autograd
        accumulates into it in place, so a step is the collective alone — no gather, no copy back.  Clear gradients
        with ``zero()`` (or ``zero_grad(set_to_none=False)``), never by setting them to None;
      * ``attach_hooks()``: the collective is launched asynchronously from the backward itself, the moment the last
        gradient of the bucket has been accumulated (it then overlaps whatever the backward still has to do for
        parameters outside the bucket, and the host's return from ``backward``); ``finish()`` waits for it.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], grads_as_views: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self._work = None
        self._pending_div = None
        self._hooks = []
        self._arrived = 0
        self._launched = False          # a collective has been started (by start() or by the hooks) and not finished
        if grads_as_views:
            self.install_views()

    def nbytes(self) -> int:
        return self.flat.numel() * 4

    # ---- gradients as views of the flat buffer ---------------------------------------------------------------
    def _views(self):
        return [v.view(p.shape) for v, p in zip(self.flat.split(self.sizes), self.params)]

    def install_views(self) -> None:
        """Make every ``.grad`` a slice of the flat buffer (current gradient values are kept)."""
        for p, v in zip(self.params, self._views()):
            if p.dtype != torch.float32:
                raise ValueError("gradient views need fp32 parameters")
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            elif p.grad is None:
                v.zero_()
            p.grad = v

    def _is_viewed(self) -> bool:
        off = 0
        base = self.flat.data_ptr()
        for p, n in zip(self.params, self.sizes):
            if p.grad is None or p.grad.dtype != torch.float32 or p.grad.data_ptr() != base + 4 * off or not p.grad.is_contiguous():
                return False
            off += n
        return True

    def zero(self) -> None:
        """One launch: clear every gradient of the bucket (gradients as views)."""
        self.flat.zero_()

    # ---- the collective --------------------------------------------------------------------------------------
    def start(self, average: bool = True, group=None) -> None:
        """Gather (unless the gradients are views) and launch the all-reduce without waiting for it."""
        if self._launched:
            raise RuntimeError("GradBucket.start(): the previous collective has not been finished — call finish() once per "
                               "backward (gradient accumulation over several backwards needs detach_hooks() + allreduce())")
        self._arrived = 0
        self._launched = True
        self._scatter = not self._is_viewed()
        if self._scatter:
            missing = [i for i, p in enumerate(self.params) if p.grad is None]
            for i in missing:
                self.params[i].grad = torch.zeros_like(self.params[i], dtype=torch.float32)
            grads = [p.grad for p in self.params]
            if all(g.dtype == torch.float32 and g.is_contiguous() for g in grads):
                torch.cat([g.reshape(-1) for g in grads], out=self.flat)           # one launch
            else:
                for v, g in zip(self.flat.split(self.sizes), grads):
                    v.copy_(g.reshape(-1))
        self._work, self._pending_div = None, None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world = dist.get_world_size(group)
            op = dist.ReduceOp.SUM
            if average:
                if dist.get_backend(group) == "nccl":
                    op = dist.ReduceOp.AVG                    # the division inside the collective: no extra launch
                else:
                    self._pending_div = world
            try:
                self._work = dist.all_reduce(self.flat, op=op, group=group, async_op=True)
            except (RuntimeError, ValueError):
                if op == dist.ReduceOp.SUM:
                    raise
                self._pending_div = world                     # a backend build without AVG: sum, divide afterwards
                self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def finish(self) -> None:
        """Wait for the collective launched by ``start`` (or by the hooks) and put the result where ``.grad`` is.

        With hooks attached, a backward in which some parameter of the bucket received no gradient never launches the
        collective: that is an error here (in a multi-rank job the ranks' parameters would silently drift apart)."""
        if self._hooks and not self._launched:
            arrived, self._arrived = self._arrived, 0
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            if multi or arrived:
                raise RuntimeError(f"GradBucket.finish(): no all-reduce was launched by the backward hooks "
                                   f"({arrived} of {len(self.params)} gradients arrived since the last finish()); every "
                                   f"parameter of the bucket must receive a gradient in every backward")
        self._launched = False
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._pending_div is not None:
            self.flat.div_(self._pending_div)
            self._pending_div = None
        if getattr(self, "_scatter", False):
            grads = [p.grad for p in self.params]
            shaped = self._views()
            if all(g.dtype == torch.float32 for g in grads):
                torch._foreach_copy_(grads, shaped)                               # one launch (fused foreach)
            else:
                for g, v in zip(grads, shaped):
                    g.copy_(v)
            self._scatter = False
        self._arrived = 0

    def allreduce(self, average: bool = True, group=None) -> None:
        """Blocking form: ``start`` + ``finish``."""
        self.start(average, group)
        self.finish()

    # ---- launched from the backward ----------------------------------------------------------------------------
    def attach_hooks(self, average: bool = True, group=None) -> None:
        """Launch the all-reduce from inside ``backward``: a post-accumulate hook on every parameter counts arrivals and
        the last one calls ``start``.  Call ``finish()`` after ``backward`` (before the optimiser step).  Every
        parameter of the bucket must receive a gradient in every backward."""
        self.detach_hooks()
        n = len(self.params)

        def hook(_p):
            if self._launched:
                raise RuntimeError("GradBucket: a gradient arrived while the all-reduce of the previous backward is still "
                                   "pending — call finish() after every backward")
            self._arrived += 1
            if self._arrived == n:
                self.start(average, group)          # resets the arrival counter
        self._hooks = [p.register_post_accumulate_grad_hook(hook) for p in self.params]

    def detach_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._arrived = 0
        if self._work is None:
            self._launched = False
