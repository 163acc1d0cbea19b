"""hipGraph capture of PDE-layer steps.

The layers of the reference's own models (C = 1 or 3 channels, batch 64-128) finish on the device in tens of
microseconds; an eager forward + backward is bound by the host's launch path (Python, autograd, ctypes: ~0.2-0.5 ms for
the three cifar10 layers against ~0.2 ms of device time).  With an explicit checkpoint plan the library's calls are
launches only — no wait for the coefficient maxima, no host copy — so a whole step can be captured once and replayed.

    freeze_checkpoint_plans(module)                       every PDE layer's plan pinned to masks (from its parameters alone)
    make_graphed(module, *sample_args)                    torch.cuda.make_graphed_callables on top of that: forward and
                                                          backward replay graphs, usable like the module itself
    GraphedStep(fn, warmup=3)                             a whole ``grads = fn()`` (forward + autograd.grad) as one graph

The plans are frozen at capture time from the parameters of that moment (lagged-policy budget: half the error budget as
margin); after large parameter changes call ``freeze_checkpoint_plans`` and capture again."""
from __future__ import annotations

import gc

import torch

from . import layers as _layers

__all__ = ["freeze_checkpoint_plans", "make_graphed", "GraphedStep"]


def _pde_layers(module):
    return [m for m in module.modules() if isinstance(m, _layers._AdiBase)]


def freeze_checkpoint_plans(module, *example_inputs):
    """Pin every implicit PDE layer's checkpoint plan to explicit masks made from its CURRENT parameters
    (``layer.freeze_checkpoint_plan()``: one small kernel per layer, no forward pass of the module — BatchNorm
    statistics, dropout and every other side effect of a forward stay untouched, and a layer the example pass would not
    have reached is planned like the others).  ``example_inputs`` is accepted for compatibility and ignored.
    Returns {layer: mask or tuple of masks}."""
    return {ly: ly.freeze_checkpoint_plan() for ly in _pde_layers(module)}


def make_graphed(module, *sample_args, num_warmup_iters=3):
    """``torch.cuda.make_graphed_callables`` for a module built from the PDE layers: plans frozen first, then the
    forward and the backward are captured (one graph each) and replayed on every call.  Shapes, dtypes and the
    requires_grad pattern of the arguments are those of ``sample_args`` from then on."""
    freeze_checkpoint_plans(module, *sample_args)
    return torch.cuda.make_graphed_callables(module, tuple(sample_args), num_warmup_iters=num_warmup_iters)


class GraphedStep:
    """``fn()`` — typically a forward plus ``torch.autograd.grad`` over fixed input tensors — captured once and
    replayed: ``outputs = step()`` returns the SAME output tensors every time (static buffers; copy what must outlive
    the next replay).  The tensors ``fn`` reads must be updated in place between replays.  Do not keep an autograd graph
    over the same leaves alive while capturing (e.g. an earlier output with a ``grad_fn``): torch then synchronises the
    capture with the stream that graph was built on, which a capture does not survive (torch's AccumulateGrad warning)."""

    def __init__(self, fn, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # No cyclic garbage collection while the stream is capturing: an unreachable CUDAGraph of an earlier step (torch's
        # graphed callables sit in reference cycles) destroyed from inside the capture — e.g. by a collection that a
        # backward running in the autograd thread happens to trigger — is "operation not permitted when stream is
        # capturing" raised from a destructor, i.e. the process aborts.
        gc.collect()
        was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                outs = fn()
        finally:
            if was_on:
                gc.enable()
        # Keep the static buffers, not the autograd graph of the capture: an output with a grad_fn would keep the capture
        # stream's AccumulateGrad nodes of the leaves alive, and the next eager backward over the same leaves (on another
        # stream) would trip over them ("AccumulateGrad node's stream does not match ...").
        det = lambda o: o.detach() if torch.is_tensor(o) else o
        self.outputs = type(outs)(det(o) for o in outs) if isinstance(outs, (tuple, list)) else det(outs)
        del outs

    def __call__(self):
        self.graph.replay()
        return self.outputs
