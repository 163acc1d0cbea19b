"""GPU parity, gate 4: the whole-layer kernels for C <= 4 channels (pde_adi_small_*: one launch per pass, the C x C
operator applied in registers at the step boundaries, skip blend fused) — the shapes the reference's own models run
(cifar10.py:253-258, SVHN.py:238, cifar_2version.py:269-270).  Against the CPU oracle at 1e-5 and against the
per-step launch path of the same layer.  The golden vectors cifar10_* / svhn_* / cifar2_* of test_gpu_golden.py take
this path as well."""
import contextlib
import copy
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


def _randomise(layer, g, rel=0.2, slope=0.3, live_matrix=0.1):
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + rel * torch.randn(p.shape, generator=g))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(slope * torch.randn(p.shape, generator=g))
            elif n in ("channel_mixing", "channel_coupling"):
                C = p.shape[0]
                p.copy_(torch.eye(C) + live_matrix * torch.randn(C, C, generator=g))
            elif n == "skip_weight":
                p.fill_(0.3)


def _run(layer, u, gy, dtype=torch.float32):
    dl = layer.cuda()
    for p in dl.parameters():
        p.grad = None
    ud = u.to(dtype).cuda().requires_grad_(True)
    y = dl(ud)
    y.backward(gy.to(dtype).cuda())
    torch.cuda.synchronize()
    out = {"y": y.detach().float().cpu(), "gu": ud.grad.float().cpu()}
    for n, p in dl.named_parameters():
        out["g_" + n] = p.grad.float().cpu()
    return out


def _check(got, ref, tol, what=""):
    errs = {k: G.rel_err(got[k].reshape(ref[k].shape), ref[k]) for k in ref}
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, (what, bad, errs)


def _oracle(layer, spec, u, gy, state_cast=None):
    params = {k: v.detach().cpu().clone() for k, v in layer.named_parameters()}
    y, gu, gp = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec, state_cast), u, params, gy)
    ref = {"y": y, "gu": gu}
    ref.update({"g_" + k: v for k, v in gp.items()})
    return ref


CASES = [
    # kind, C, N, steps, dt, B
    ("cifar10", 3, 32, 5, 0.02, 9),          # the reference's shape: mixing before every Strang step
    ("cifar10", 1, 32, 2, 0.05, 3),          # a 1 x 1 operator
    ("cifar10", 4, 28, 3, 0.03, 6),          # idle lanes (N < 32), four waves
    ("cifar10", 2, 16, 4, 0.05, 5),
    ("cifar2", 3, 32, 6, 0.02, 7),           # Lie steps
    ("cifar2", 4, 16, 3, 0.05, 2),
    ("svhn", 3, 32, 10, 0.01, 8),            # the reference's shape: coupling after every step, skip blend, smoothing
    ("svhn", 2, 28, 3, 0.05, 4),
    ("svhn", 4, 32, 2, 0.02, 1),             # a single sample
]


def _make(kind, C, N, steps, dt):
    import cnn_with_pde_amd as P
    if kind == "cifar10":
        return quiet(P.EnhancedDiffusionLayer, N, C, dt=dt, num_steps=steps), O.cifar10_spec(N, C, dt=dt, num_steps=steps)
    if kind == "cifar2":
        return quiet(P.LearnableDiffusionLayer, N, C, dt=dt, num_steps=steps), O.cifar2_spec(N, C, dt=dt, num_steps=steps)
    return P.SvhnDiffusionLayer(N, C, dt=dt, dx=1.0, num_steps=steps), O.svhn_spec(N, C, dt=dt, dx=1.0, num_steps=steps)


@pytest.mark.parametrize("kind,C,N,steps,dt,B", CASES)
def test_small_kernels_vs_oracle_and_per_step_path(kind, C, N, steps, dt, B):
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(900 + 7 * C + N + steps)
    layer, spec = _make(kind, C, N, steps, dt)
    _randomise(layer, g)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    assert F_.adi_small_supported(u.cuda(), layer._schedule(), smooth3=layer._smooth3, clamp_max=layer._clamp_max)
    ref = _oracle(layer, spec, u, gy)
    got = _run(copy.deepcopy(layer), u, gy)
    _check(got, ref, TOL, "single launch vs oracle")
    other = copy.deepcopy(layer)
    other.small_channel_kernels = False                  # the per-step launch path of the same layer
    got2 = _run(other, u, gy)
    _check(got, got2, 5e-6, "single launch vs per-step launches")


def test_small_kernels_time_varying_clamp_mask():
    """Coefficients that cross the clamp bounds during the time window (cifar10.py:60-61: [1e-6, 10]): the channel
    takes the per-sweep mask path inside the same launch; the other channels stay on the fast path."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(31)
    C, N, steps, dt, B = 3, 32, 4, 0.25, 5
    layer = quiet(P.EnhancedDiffusionLayer, N, C, dt=dt, num_steps=steps)
    _randomise(layer, g, 0.2, 0.0)
    with torch.no_grad():
        layer.alpha_base[1].fill_(9.9)                   # channel 1 crosses the upper bound half-way ...
        layer.alpha_time_coeff[1].copy_(0.4 + 0.1 * torch.randn(N, N, generator=g))
        layer.beta_base[1, :8].fill_(0.02)               # ... and the lower one in a few rows
        layer.beta_time_coeff[1, :8].fill_(-0.05)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    ref = _oracle(layer, O.cifar10_spec(N, C, dt=dt, num_steps=steps), u, gy)
    _check(_run(layer, u, gy), ref, TOL)


@pytest.mark.parametrize("ck", ["auto", 0b01, 0b11])
def test_small_kernels_large_coefficients_checkpoints(ck):
    """fashion-size coefficients (0.54) on the SVHN layer at C = 3: the backward re-runs the forward to park states
    inside the steps (automatic plan and explicit masks)."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(77)
    C, N, B = 3, 28, 6
    layer = P.SvhnDiffusionLayer(N, C, dt=0.3, dx=1.0, num_steps=4)
    layer.checkpoint_policy = ck
    with torch.no_grad():
        layer.alpha_base.copy_(1.8 * (1 + 0.1 * torch.randn(C, N, N, generator=g)))
        layer.beta_base.copy_(1.8 * (1 + 0.1 * torch.randn(C, N, N, generator=g)))
        layer.alpha_time_coeff.copy_(0.2 * torch.randn(C, N, N, generator=g))
        layer.beta_time_coeff.copy_(0.2 * torch.randn(C, N, N, generator=g))
        layer.channel_coupling.copy_(torch.eye(C) + 0.1 * torch.randn(C, C, generator=g))
        layer.skip_weight.fill_(-0.4)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    ref = _oracle(layer, O.svhn_spec(N, C, dt=0.3, dx=1.0, num_steps=4), u, gy)
    _check(_run(layer, u, gy), ref, TOL)


def test_small_kernels_bf16_and_many_samples():
    """bf16 tensors (fp32 arithmetic inside; against the oracle rounding its state where the kernel does), and a batch
    larger than the grid (samples are walked with a grid stride; partial sums per workgroup)."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(5)
    layer, spec = _make("svhn", 3, 32, 3, 0.02)
    _randomise(layer, g)
    u = torch.randn(5, 3, 32, 32, generator=g).bfloat16().float()
    gy = torch.randn(5, 3, 32, 32, generator=g).bfloat16().float()
    ref = _oracle(layer, spec, u, gy)
    _check(_run(copy.deepcopy(layer), u, gy, torch.bfloat16), ref, 2e-2)
    # additivity over a batch of 2500 samples > 1024 workgroups: two halves add up to the whole
    layer2, _ = _make("cifar10", 3, 16, 2, 0.05)
    _randomise(layer2, g)
    ub = torch.randn(2500, 3, 16, 16, generator=g)
    gb = torch.randn(2500, 3, 16, 16, generator=g)
    whole = _run(copy.deepcopy(layer2), ub, gb)
    h1 = _run(copy.deepcopy(layer2), ub[:1100], gb[:1100])
    h2 = _run(copy.deepcopy(layer2), ub[1100:], gb[1100:])
    assert G.rel_err(torch.cat([h1["y"], h2["y"]]), whole["y"]) <= 1e-6
    assert G.rel_err(torch.cat([h1["gu"], h2["gu"]]), whole["gu"]) <= 1e-6
    for k in whole:
        if k.startswith("g_"):
            assert G.rel_err(h1[k] + h2[k], whole[k]) <= TOL, k
    ref2 = _oracle(layer2, O.cifar10_spec(16, 3, dt=0.05, num_steps=2), ub[:40], gb[:40])
    _check(_run(copy.deepcopy(layer2), ub[:40], gb[:40]), ref2, TOL)


def test_small_kernels_inference_and_retained_graph():
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(12)
    layer, spec = _make("svhn", 3, 32, 4, 0.02)
    _randomise(layer, g)
    layer = layer.cuda()
    u = torch.randn(4, 3, 32, 32, generator=g).cuda()
    gy = torch.randn(4, 3, 32, 32, generator=g).cuda()
    with torch.no_grad():
        y0 = layer(u)                                    # no states kept
    ud = u.clone().requires_grad_(True)
    y = layer(ud)
    assert torch.equal(y0, y.detach())
    y.backward(gy, retain_graph=True)
    first = {n: p.grad.clone() for n, p in layer.named_parameters()}
    for p in layer.parameters():
        p.grad = None
    ud.grad = None
    y.backward(gy)
    for n, p in layer.named_parameters():
        assert torch.equal(p.grad, first[n]), n
