"""hipGraph capture of PDE-layer steps.

The layers of the reference's own models (C = 1 or 3 channels, batch 64-128) finish on the device in tens of
microseconds; an eager forward + backward is bound by the host's launch path (Python, autograd, ctypes: ~0.2-0.5 ms for
the three cifar10 layers against ~0.2 ms of device time).  With an explicit checkpoint plan the library's calls are
launches only — no wait for the coefficient maxima, no host copy — so a whole step can be captured once and replayed.

    freeze_checkpoint_plans(module, *example_inputs)      every PDE layer's plan pinned to a mask (one eager pass)
    make_graphed(module, *sample_args)                    torch.cuda.make_graphed_callables on top of that: forward and
                                                          backward replay graphs, usable like the module itself
    GraphedStep(fn, inputs)                               a whole ``grads = fn()`` (forward + autograd.grad) as one graph

The plans are frozen at capture time from the parameters of that moment (lagged-policy budget: half the error budget as
margin); after large parameter changes call ``freeze_checkpoint_plans`` and capture again."""
from __future__ import annotations

import torch

from . import layers as _layers

__all__ = ["freeze_checkpoint_plans", "make_graphed", "GraphedStep"]


def _pde_layers(module):
    return [m for m in module.modules() if isinstance(m, _layers._AdiBase)]


def freeze_checkpoint_plans(module, *example_inputs):
    """Run ``module(*example_inputs)`` once eagerly with every implicit PDE layer recording the checkpoint plan its
    current coefficients need, and pin those plans as explicit masks.  Returns {layer: mask}."""
    found = _pde_layers(module)
    for ly in found:
        ly.__dict__.pop("_kmax_cache", None)
        ly.__dict__["_plans_seen"] = []
        ly.__dict__["_policy_before"] = ly.checkpoint_policy
        ly.checkpoint_policy = "lagged"
    try:
        with torch.enable_grad():
            args = [a.detach().clone().requires_grad_(True) if torch.is_tensor(a) and a.is_floating_point() else a
                    for a in example_inputs]
            module(*args)
    finally:
        for ly in found:
            ly.checkpoint_policy = ly.__dict__.pop("_policy_before")
    out = {}
    for ly in found:
        mask = 0
        for m in ly.__dict__.pop("_plans_seen", []):
            mask |= int(m)
        ly.__dict__.pop("_kmax_cache", None)
        ly.checkpoint_policy = mask
        out[ly] = mask
    return out


def make_graphed(module, *sample_args, num_warmup_iters=3):
    """``torch.cuda.make_graphed_callables`` for a module built from the PDE layers: plans frozen first, then the
    forward and the backward are captured (one graph each) and replayed on every call.  Shapes, dtypes and the
    requires_grad pattern of the arguments are those of ``sample_args`` from then on."""
    freeze_checkpoint_plans(module, *sample_args)
    return torch.cuda.make_graphed_callables(module, tuple(sample_args), num_warmup_iters=num_warmup_iters)


class GraphedStep:
    """``fn()`` — typically a forward plus ``torch.autograd.grad`` over fixed input tensors — captured once and
    replayed: ``outputs = step()`` returns the SAME output tensors every time (static buffers; copy what must outlive
    the next replay).  The tensors ``fn`` reads must be updated in place between replays."""

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
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs
