"""GPU: hipGraph capture of PDE-layer steps (cnn_with_pde_amd.graphs) — frozen checkpoint plans, a whole
forward + backward replayed from one graph, torch's graphed callables on top of a layer."""
import contextlib
import copy
import io

import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _grads(layer, u, gy):
    for p in layer.parameters():
        p.grad = None
    ud = u.clone().requires_grad_(True)
    y = layer(ud)
    y.backward(gy)
    return [y.detach().clone(), ud.grad.clone()] + [p.grad.clone() for p in layer.parameters()]


def test_frozen_plan_matches_the_automatic_one():
    """fashion-size coefficients: the backward must park states; the frozen mask is not empty and gives the gradients
    of the automatic plan; tiny coefficients freeze to the empty plan."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(3)
    layer = P.FashionDiffusionLayer().cuda()
    u = torch.randn(16, 1, 28, 28, generator=g).cuda()
    gy = torch.randn(16, 1, 28, 28, generator=g).cuda()
    ref = _grads(layer, u, gy)                                  # "auto"
    frozen = copy.deepcopy(layer)
    mask = frozen.freeze_checkpoint_plan(u)
    assert mask != 0 and frozen.checkpoint_policy == mask
    got = _grads(frozen, u, gy)
    for a, b in zip(got, ref):
        assert G.rel_err(a.cpu(), b.cpu()) <= 1e-5
    small = quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.001, num_steps=5).cuda()
    assert small.freeze_checkpoint_plan(torch.randn(2, 3, 32, 32, device="cuda")) == 0
    model = quiet(P.CIFAR10PDENoConv).cuda()
    plans = P.freeze_checkpoint_plans(model, torch.randn(4, 3, 32, 32, device="cuda"))
    assert len(plans) == 3 and all(v == 0 for v in plans.values())


def test_graphed_step_of_the_shared_input_layers():
    """The three cifar10 layers (one launch per pass) forward + backward as ONE graph: bitwise the eager result, and
    again after the inputs changed in place."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(11)
    layers = [quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).cuda(),
              quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(),
              quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).cuda()]
    x = torch.randn(32, 3, 32, 32, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(32, 3, 32, 32, generator=g).cuda()
    w = torch.tensor([0.5, 0.3, 0.2], device="cuda", requires_grad=True)
    for ly in layers:
        ly.freeze_checkpoint_plan(x)
    params = [p for ly in layers for p in ly.parameters()]

    def fn():
        out, _ = P.diffuse_shared_input(layers, x, w)
        return (out,) + torch.autograd.grad(out, [x, w] + params, gy)

    step = P.GraphedStep(fn)
    for trial in range(2):
        eager = [t.clone() for t in fn()]
        got = step()
        torch.cuda.synchronize()
        for a, b in zip(got, eager):
            assert torch.equal(a, b)
        with torch.no_grad():                                   # new data, same buffers
            x.copy_(torch.randn(32, 3, 32, 32, generator=g))
            gy.copy_(torch.randn(32, 3, 32, 32, generator=g))
            layers[1].alpha_base.mul_(1.1)


# (torch.cuda.make_graphed_callables warms the module up on a side stream and keeps those AccumulateGrad nodes of the
# parameters alive inside its own graphed autograd function: the stream-mismatch warning of the eager backward below is
# torch's, about torch's nodes — this package's GraphedStep keeps no autograd graph, see graphs.py)
@pytest.mark.filterwarnings("ignore:The AccumulateGrad node's stream does not match")
def test_make_graphed_layer_trains_like_the_eager_one():
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(5)
    layer = P.SvhnDiffusionLayer(32, 3, dt=0.01, num_steps=4).cuda()
    with torch.no_grad():
        layer.channel_coupling.copy_((torch.eye(3) + 0.05 * torch.randn(3, 3, generator=g)).cuda())
    eager = copy.deepcopy(layer)
    x = torch.randn(8, 3, 32, 32, generator=g).cuda()
    gy = torch.randn(8, 3, 32, 32, generator=g).cuda()
    graphed = P.make_graphed(layer, x.clone().requires_grad_(True))
    for _ in range(2):
        xa = x.clone().requires_grad_(True)
        ya = graphed(xa)
        ya.backward(gy)
        ref = _grads(eager, x, gy)
        assert G.rel_err(ya.detach().cpu(), ref[0].cpu()) <= 1e-6
        assert G.rel_err(xa.grad.cpu(), ref[1].cpu()) <= 1e-6
        for p, r in zip(layer.parameters(), ref[2:]):
            assert G.rel_err(p.grad.cpu(), r.cpu()) <= 1e-6
        for p in layer.parameters():
            p.grad = None
        x = x * 0.5 + 0.1


@pytest.mark.parametrize("kind", ["cifar10_c32", "svhn_c32_checkpointed"])
def test_graphed_step_of_a_wide_layer(kind):
    """Layers with a channel operator at C = 32 (one-launch forward, per-step backward; with fashion-size coefficients
    the backward parks states and recomputes the operator outputs the forward did not keep): captured and replayed
    bitwise like the eager call."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(19)
    if kind == "cifar10_c32":
        layer = quiet(P.EnhancedDiffusionLayer, 28, 32, dt=0.02, num_steps=3).cuda()
    else:
        layer = P.SvhnDiffusionLayer(28, 32, dt=0.3, num_steps=2).cuda()
        with torch.no_grad():
            layer.alpha_base.fill_(1.8)
            layer.beta_base.fill_(1.8)
            layer.channel_coupling.copy_((torch.eye(32) + 0.05 * torch.randn(32, 32, generator=g)).cuda())
    x = torch.randn(6, 32, 28, 28, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(6, 32, 28, 28, generator=g).cuda()
    mask = layer.freeze_checkpoint_plan(x)
    assert (mask != 0) == (kind == "svhn_c32_checkpointed")
    params = list(layer.parameters())

    def fn():
        y = layer(x)
        return (y,) + torch.autograd.grad(y, [x] + params, gy)

    step = P.GraphedStep(fn)
    eager = [t.clone() for t in fn()]
    got = step()
    torch.cuda.synchronize()
    for a, b in zip(got, eager):
        assert torch.equal(a, b)


def test_freezing_plans_has_no_side_effects():
    """The plans come from the parameters alone: no forward pass of the model, so BatchNorm running statistics and
    num_batches_tracked stay where they are (graphs.py ran an eager training-mode forward)."""
    import cnn_with_pde_amd as P
    model = quiet(P.CIFAR10PDENoConv).cuda().train()
    before = {n: b.clone() for n, b in model.named_buffers()}
    state = torch.cuda.get_rng_state()
    P.freeze_checkpoint_plans(model)
    assert torch.equal(state, torch.cuda.get_rng_state())
    for n, b in model.named_buffers():
        assert torch.equal(b, before[n]), n


def test_shared_input_layers_with_different_frozen_masks_stay_fused():
    """Three mixing-first layers on one input whose frozen masks DIFFER (different coefficient sizes): still one launch per
    pass (per-layer masks reach pde_adi_multi_*), capturable, and bitwise the automatic plan's result."""
    import cnn_with_pde_amd as P
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(21)
    layers = [quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.001, num_steps=4).cuda(),
              quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.6, num_steps=4).cuda(),
              quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.6, num_steps=4).cuda()]
    with torch.no_grad():
        layers[2].alpha_base.mul_(6.0)                           # larger coefficients along x only: another plan
    x = torch.randn(5, 3, 32, 32, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(5, 3, 32, 32, generator=g).cuda()
    w = torch.tensor([0.5, 0.3, 0.2], device="cuda", requires_grad=True)
    params = [p for ly in layers for p in ly.parameters()]

    def fn():
        out, _ = P.diffuse_shared_input(layers, x, w)
        return (out,) + torch.autograd.grad(out, [x, w] + params, gy)
    auto = [t.detach().clone() for t in fn()]            # (no grad_fn kept: a live autograd graph of the default stream
                                                         #  would drag that stream into the capture below — torch's warning)
    masks = [ly.freeze_checkpoint_plan() for ly in layers]
    assert masks[0] == 0 and masks[1] != 0 and masks[2] != masks[1], masks
    calls = []
    orig = F_.adi_diffuse_multi
    F_.adi_diffuse_multi = lambda *a, **k: (calls.append(a[4] if len(a) > 4 else k.get("checkpoints")), orig(*a, **k))[1]
    try:
        step = P.GraphedStep(fn)
    finally:
        F_.adi_diffuse_multi = orig
    assert calls and all(c == tuple(masks) for c in calls), calls      # the fused entry point, with one mask per layer
    got = step()
    torch.cuda.synchronize()
    for a, b in zip(got, auto):
        assert G.rel_err(a.cpu(), b.cpu()) <= 1e-5


def test_long_schedule_freezes_one_mask_per_launch_group():
    """num_steps > 32 Strang steps run as several launch groups (PDE_MAX_SWEEPS = 96): a frozen plan is one mask per group
    (the tail group is shorter: a single union mask would carry bits beyond its schedule), and a single int mask given
    by hand is cut to each group's own bits."""
    import cnn_with_pde_amd as P
    from oracle import pde_oracle as O
    g = torch.Generator().manual_seed(4)
    layer = quiet(P.MnistDiffusionLayer, 12, dt=0.2, dx=1.0, dy=1.0, num_steps=35)
    with torch.no_grad():
        layer.alpha_base.mul_(0.7 + 0.3 * torch.rand(12, 12, generator=g))
    u = torch.randn(3, 1, 12, 12, generator=g)
    gy = torch.randn(3, 1, 12, 12, generator=g)
    spec = O.mnist_spec(12, 0.2, 1.0, 1.0, 35)
    params = {k: v.detach().clone() for k, v in layer.named_parameters()}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    dl = layer.cuda()
    masks = dl.freeze_checkpoint_plan()
    assert isinstance(masks, tuple) and len(masks) == 2 and masks[0] != 0 and masks[1] != 0
    assert masks[1] < (1 << 8)                                   # the tail group has 3 steps = 9 sweeps: bits 0..7
    for policy in (masks, "auto", "lagged", masks[0]):           # masks[0] (32-step plan) on both groups: cut to the tail's bits
        dl.checkpoint_policy = policy
        got = _grads(dl, u.cuda(), gy.cuda())
        errs = [G.rel_err(got[0].cpu(), y_ref), G.rel_err(got[1].cpu(), gu_ref)]
        errs += [G.rel_err(a.cpu(), gp_ref[n]) for a, (n, _) in zip(got[2:], dl.named_parameters())]
        assert max(errs) <= 1e-5, (policy, errs)


def test_more_outstanding_forwards_than_ring_entries():
    """More PDE-layer forwards outstanding than the coefficient-maxima ring has entries (functional.KMAX_RING): the extra
    calls get buffers of their own instead of raising in the backward."""
    import cnn_with_pde_amd as P
    from cnn_with_pde_amd import functional as F_
    layer = quiet(P.EnhancedDiffusionLayer, 8, 2, dt=0.01, num_steps=1, channel_mixing_enabled=False).cuda()
    u = torch.randn(1, 2, 8, 8, device="cuda", requires_grad=True)
    outs = [layer(u) for _ in range(F_.KMAX_RING + 40)]
    torch.stack([o.sum() for o in outs]).sum().backward()
    torch.cuda.synchronize()
    one = torch.autograd.grad(layer(u).sum(), u)[0]
    assert G.rel_err(u.grad.cpu(), (one * len(outs)).cpu()) <= 1e-5


def test_shared_input_backward_notices_in_place_changes():
    """The fused shared-input call saves its input and parameters through save_for_backward: changing a parameter in place
    between forward and backward raises instead of mixing the forward's factorisation with new values."""
    import cnn_with_pde_amd as P
    layers = [quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.001, num_steps=2).cuda() for _ in range(2)]
    x = torch.randn(2, 3, 32, 32, device="cuda", requires_grad=True)
    out, _ = P.diffuse_shared_input(layers, x, torch.tensor([0.5, 0.5], device="cuda"))
    with torch.no_grad():
        layers[0].alpha_base.mul_(1.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()


CYCLE_CHILD = r"""
import contextlib, gc, io, sys, weakref
sys.path[:0] = [%(root)r, %(tests)r]
import torch
import cnn_with_pde_amd as P

def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)

g = torch.Generator().manual_seed(3)
x = torch.randn(4, 3, 32, 32, generator=g).cuda()
# 1. a graphed callable that only a reference cycle keeps alive (what torch.cuda.make_graphed_callables leaves behind
#    once its caller drops it), collectable but not yet collected when the next capture starts
gc.disable()
layer0 = quiet(P.SvhnDiffusionLayer, 32, 3, dt=0.01, num_steps=2).cuda()
graphed = P.make_graphed(layer0, x.clone().requires_grad_(True))
graphed(x.clone().requires_grad_(True)).sum().backward()
class Box: pass
box = Box(); box.self = box; box.fn = graphed
probe = weakref.ref(box)
del box, graphed
assert probe() is not None            # unreachable, still there
gc.enable()
# 2. a capture with a backward inside (the autograd worker runs it), under the guard
layer = quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.6, num_steps=3).cuda()     # large coefficients: a checkpointed backward
xs = x.clone().requires_grad_(True)
gy = torch.randn(4, 3, 32, 32, generator=g).cuda()
params = list(layer.parameters())
def fn():
    return torch.autograd.grad(layer(xs), [xs] + params, gy)
ref = [t.detach().clone() for t in fn()]
layer.freeze_checkpoint_plan(xs)
step = P.GraphedStep(fn)
assert probe() is None                # the guard collected the cycle BEFORE the stream started capturing
for _ in range(3):
    got = step()
torch.cuda.synchronize()
for a, b in zip(got, ref):          # (the frozen plan may park other states than the automatic one: rounding-level differences)
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), float((a - b).abs().max())
# 3. the same around make_graphed
box = Box(); box.self = box; box.fn = step
probe = weakref.ref(box)
del box, step
g2 = P.make_graphed(layer, x.clone().requires_grad_(True))
assert probe() is None
g2(x.clone().requires_grad_(True)).sum().backward()
torch.cuda.synchronize()
print("capture guard ok")
"""


def test_capture_with_cyclic_garbage_around():
    """The conditions the capture guard of cnn_with_pde_amd.graphs exists for (see _capture_guard): an unreachable
    graphed callable in a reference cycle when a capture starts, and a backward inside the capture.  In a child
    interpreter: a failure here is a dead process, not an exception."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = CYCLE_CHILD % {"root": os.path.dirname(here), "tests": here}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "capture guard ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
