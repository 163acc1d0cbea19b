"""GPU parity, gate 5: the one-launch forward for C = 32 / 64 fp32 channels with a channel operator between the time
steps (pde_adi_wide.h: one workgroup owns all channels of a sample, MFMA mixing through an LDS exchange image,
lane-major coefficient records) — cifar10.py:84-112 (mixing before every Strang step), cifar_2version.py:77-103 (Lie
steps), SVHN.py:55-76 (coupling after every step, skip blend).  Against the CPU oracle at 1e-5, and against the
same layer at other batch sizes on a batch larger than the grid.  (tests/test_gpu_configs.py holds the mixed
layer call to its per-step composition — launches that never take this path — at C = 32 ... 128.)"""
import contextlib
import copy
import ctypes as C
import io

import pytest
import torch

import golden_util as G
from oracle import pde_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _randomise(layer, g):
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif n in ("channel_mixing", "channel_coupling"):
                c = p.shape[0]
                p.copy_(torch.eye(c) + (0.4 / c ** 0.5) * torch.randn(c, c, generator=g))
            elif n == "skip_weight":
                p.fill_(0.3)


def _make(kind, C_, N, steps, dt):
    import cnn_with_pde_amd as P
    if kind == "cifar10":
        return quiet(P.EnhancedDiffusionLayer, N, C_, dt=dt, num_steps=steps), O.cifar10_spec(N, C_, dt=dt, num_steps=steps)
    if kind == "cifar2":
        return quiet(P.LearnableDiffusionLayer, N, C_, dt=dt, num_steps=steps), O.cifar2_spec(N, C_, dt=dt, num_steps=steps)
    return P.SvhnDiffusionLayer(N, C_, dt=dt, dx=1.0, num_steps=steps), O.svhn_spec(N, C_, dt=dt, dx=1.0, num_steps=steps)


CASES = [
    # kind, C, N, steps, dt, B
    ("cifar10", 64, 32, 3, 0.02, 3),
    ("cifar10", 32, 28, 2, 0.05, 5),          # idle lanes, two elements of padding per half row
    ("cifar2", 64, 28, 3, 0.03, 2),           # Lie steps
    ("cifar2", 32, 32, 4, 0.02, 3),
    ("svhn", 64, 32, 3, 0.02, 2),             # coupling after the step, skip blend
    ("svhn", 32, 28, 2, 0.05, 1),             # a single sample
]


@pytest.mark.parametrize("kind,C_,N,steps,dt,B", CASES)
def test_wide_forward_layer_vs_oracle(kind, C_, N, steps, dt, B):
    from cnn_with_pde_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(4000 + C_ + N + steps)
    layer, spec = _make(kind, C_, N, steps, dt)
    _randomise(layer, g)
    u = torch.randn(B, C_, N, N, generator=g)
    gy = torch.randn(B, C_, N, N, generator=g)
    params = {k: v.detach().clone() for k, v in layer.named_parameters()}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    dl = layer.cuda()
    with torch.no_grad():
        y0 = dl(u.cuda())                                 # inference call
    ud = u.cuda().requires_grad_(True)
    y = dl(ud)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    assert torch.equal(y0, y.detach())
    errs = {"y": G.rel_err(y.detach().cpu(), y_ref), "gu": G.rel_err(ud.grad.cpu(), gu_ref)}
    for n, p in dl.named_parameters():
        errs["g_" + n] = G.rel_err(p.grad.cpu().reshape(gp_ref[n].shape), gp_ref[n])
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)
    # this shape does take the one-launch path
    from cnn_with_pde_amd import functional as F_
    sched = layer._schedule()
    d = F_._make_desc(B, C_, N, L.PDE_IO_F32, sched.flat, layer._smooth3, layer._clamp_max, layer.stability_eps)
    assert lib.pde_adi_mixed_one_launch(C.byref(d), len(sched[0])) == 1
    d16 = F_._make_desc(B, 16, N, L.PDE_IO_F32, sched.flat, layer._smooth3, layer._clamp_max, layer.stability_eps)
    assert lib.pde_adi_mixed_one_launch(C.byref(d16), len(sched[0])) == 0


def test_wide_forward_batch_larger_than_grid():
    """2100 samples > 2048 workgroups: the samples are walked with a grid stride."""
    g = torch.Generator().manual_seed(77)
    layer, _ = _make("cifar10", 32, 28, 2, 0.05)
    _randomise(layer, g)
    layer = layer.cuda()
    u = torch.randn(2100, 32, 28, 28, generator=g).cuda()
    with torch.no_grad():
        y = layer(u)
        other = copy.deepcopy(layer)
        parts = [other(u[i:i + 700]) for i in range(0, 2100, 700)]     # smaller batches: same kernel, other grid
        assert torch.equal(torch.cat(parts), y)
    # and the last 40 samples against the oracle
    spec = O.cifar10_spec(28, 32, dt=0.05, num_steps=2)
    params = {k: v.detach().cpu().clone() for k, v in layer.named_parameters()}
    y_ref = O.adi_forward(u[2060:].cpu(), params, spec)
    assert G.rel_err(y[2060:].cpu(), y_ref) <= TOL


@pytest.mark.parametrize("kind,ck", [("cifar10", 0b11), ("cifar10", "auto"), ("svhn", 0b01)])
def test_wide_forward_with_checkpointed_backward(kind, ck):
    """Coefficients large enough that the backward parks states inside the steps: it then also needs the operator
    outputs the one-launch forward did not keep and recomputes them."""
    g = torch.Generator().manual_seed(99)
    C_, N, B, steps, dt = 32, 32, 2, 3, 0.4
    layer, spec = _make(kind, C_, N, steps, dt)
    _randomise(layer, g)
    with torch.no_grad():
        layer.alpha_base.copy_(1.5 * (1 + 0.1 * torch.randn(layer.alpha_base.shape, generator=g)))
        layer.beta_base.copy_(1.5 * (1 + 0.1 * torch.randn(layer.beta_base.shape, generator=g)))
    layer.checkpoint_policy = ck
    u = torch.randn(B, C_, N, N, generator=g)
    gy = torch.randn(B, C_, N, N, generator=g)
    params = {k: v.detach().clone() for k, v in layer.named_parameters()}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    dl = layer.cuda()
    ud = u.cuda().requires_grad_(True)
    y = dl(ud)
    y.backward(gy.cuda())
    errs = {"y": G.rel_err(y.detach().cpu(), y_ref), "gu": G.rel_err(ud.grad.cpu(), gu_ref)}
    for n, p in dl.named_parameters():
        errs["g_" + n] = G.rel_err(p.grad.cpu().reshape(gp_ref[n].shape), gp_ref[n])
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)


def test_full_baseline_size_with_mixing_properties():
    """BASELINE configs[1] as the reference runs it (512 x 64 x 32 x 32, mixing before each of the 10 Strang steps):
    one-launch forward, per-step backward.  Size-independent properties: linearity in u, the adjoint identity,
    parameter gradients (channel_mixing included) of two parts of the batch add up to those of the whole, and a
    slice of the batch equals the same samples run alone."""
    g = torch.Generator().manual_seed(2025)
    layer, _ = _make("cifar10", 64, 32, 10, 0.001)
    _randomise(layer, g)
    layer = layer.cuda()
    B = 512
    u = torch.randn(B, 64, 32, 32, generator=g).cuda()
    v = torch.randn(B, 64, 32, 32, generator=g).cuda()
    gy = torch.randn(B, 64, 32, 32, generator=g).cuda()
    with torch.no_grad():
        yu, yv = layer(u), layer(v)
        lhs = layer(2.0 * u - 0.5 * v)
        rhs = 2.0 * yu - 0.5 * yv
        assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= 5e-6
        assert torch.equal(layer(u[100:132]), yu[100:132])
    del lhs, rhs, yv
    ud = u.clone().requires_grad_(True)
    y = layer(ud)
    y.backward(gy)
    a = float((y.detach().double() * gy.double()).sum())
    b = float((ud.grad.double() * u.double()).sum())
    assert abs(a - b) <= 1e-5 * float(y.detach().double().norm() * gy.double().norm())
    full = {n: p.grad.clone() for n, p in layer.named_parameters()}
    del y, ud
    acc = {n: torch.zeros_like(t) for n, t in full.items()}
    for sl in (slice(0, 200), slice(200, 512)):
        for p in layer.parameters():
            p.grad = None
        uh = u[sl].clone().requires_grad_(True)
        layer(uh).backward(gy[sl])
        for n, p in layer.named_parameters():
            acc[n] += p.grad
    for n in full:
        assert G.rel_err(acc[n].cpu(), full[n].cpu()) <= TOL, n
