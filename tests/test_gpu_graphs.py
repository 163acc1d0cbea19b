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
