"""GPU: repeated calls give bitwise identical, finite results whatever other kernels left in LDS in between — for
every supported line length (lanes beyond the plane exist for N < 32), the plain layers, the layers with a channel
operator on the per-step, one-launch (C <= 4) and shared-input paths, masked (time-varying clamp) channels included.
A sum that multiplies by a 0/1 lane factor, or an idle lane that reads LDS nobody wrote, shows up here as a NaN or as a
difference between repetitions."""
import contextlib
import io

import pytest
import torch

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _dirty():
    """Kernels that use LDS and leave arbitrary bit patterns (NaN payloads included) behind."""
    x = torch.randn(1 << 16, device="cuda")
    x[::7] = float("nan")
    x.sort()
    torch.cumsum(x, 0)
    (x.view(256, 256) @ x.view(256, 256)).sum()
    torch.softmax(x.view(64, -1), 1)


def _repeat(fn, reps=4):
    first = None
    for _ in range(reps):
        _dirty()
        cur = fn()
        for t in cur:
            assert bool(torch.isfinite(t).all()), "non-finite values"
        if first is None:
            first = cur
        else:
            for a, b in zip(cur, first):
                assert torch.equal(a, b), "repetitions differ"


def _perturb(layer, g, slope):
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=g).to(p.device))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(slope * torch.randn(p.shape, generator=g).to(p.device))
            elif n in ("channel_mixing", "channel_coupling"):
                c = p.shape[0]
                p.copy_((torch.eye(c) + 0.1 * torch.randn(c, c, generator=g)).to(p.device))


def _fwd_bwd(layer, u, gy):
    def fn():
        for p in layer.parameters():
            p.grad = None
        ud = u.clone().requires_grad_(True)
        y = layer(ud)
        y.backward(gy)
        return [y.detach(), ud.grad] + [p.grad.clone() for p in layer.parameters() if p.grad is not None]
    return fn


@pytest.mark.parametrize("N", [8, 12, 16, 20, 24, 28, 32])
@pytest.mark.parametrize("C,mixing", [(5, False), (3, True), (8, True)])
def test_layers_are_deterministic_and_finite(N, C, mixing):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(1000 + 10 * N + C)
    layer = quiet(P.EnhancedDiffusionLayer, N, C, dt=0.05, num_steps=3, channel_mixing_enabled=mixing).cuda()
    _perturb(layer, g, 25.0)                 # slopes large enough to cross the clamp bounds: masked channels
    if mixing is False:
        layer.channel_mixing.requires_grad_(False)
    B = 37
    u = torch.randn(B, C, N, N, generator=g).cuda()
    gy = torch.randn(B, C, N, N, generator=g).cuda()
    _repeat(_fwd_bwd(layer, u, gy))


@pytest.mark.parametrize("N", [16, 28, 32])
def test_shared_input_group_and_svhn_layer_are_deterministic_and_finite(N):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(77 + N)
    C = 3
    layers = [quiet(P.EnhancedDiffusionLayer, N, C, dt=dt, num_steps=st, dx=dx, dy=dx).cuda()
              for dt, st, dx in ((0.02, 2, 1.0), (0.05, 3, 2.0), (0.03, 1, 1.5))]
    for ly in layers:
        _perturb(ly, g, 0.2)
    w = torch.softmax(torch.randn(3, generator=g), 0).cuda().requires_grad_(True)
    u = torch.randn(50, C, N, N, generator=g).cuda()
    gy = torch.randn(50, C, N, N, generator=g).cuda()
    params = [p for ly in layers for p in ly.parameters()]

    def fused():
        ud = u.clone().requires_grad_(True)
        out, _ = P.diffuse_shared_input(layers, ud, w)
        return [out.detach()] + list(torch.autograd.grad(out, [ud, w] + params, gy))
    _repeat(fused)
    sv = P.SvhnDiffusionLayer(N, C, dt=0.3, num_steps=3).cuda()      # coupling after the step, skip blend, checkpoints
    _perturb(sv, g, 0.2)
    with torch.no_grad():
        sv.alpha_base.mul_(15.0)
        sv.beta_base.mul_(15.0)
    _repeat(_fwd_bwd(sv, u, gy))


@pytest.mark.parametrize("size,steps,dtype", [(16, 2, torch.float32), (32, 1, torch.bfloat16), (64, 3, torch.float32),
                                              (24, 2, torch.float32), (40, 1, torch.bfloat16)])
def test_explicit_layers_are_deterministic_and_finite(size, steps, dtype):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(5 + size)
    layer = P.ImprovedDiffusionLayer(size, 6, num_steps=steps).cuda()
    with torch.no_grad():
        layer.alpha_base.copy_(0.02 + 0.2 * torch.rand(6, generator=g))     # some above the 0.15 clamp
        layer.channel_scaling.copy_(1 + 0.2 * torch.randn(6, generator=g))
    u = torch.randn(21, 6, size, size, generator=g).to(dtype).cuda()
    gy = torch.randn(21, 6, size, size, generator=g).to(dtype).cuda()
    _repeat(_fwd_bwd(layer, u, gy))
    if size == 16:
        pl = P.PDELayer(Nx=20, Ny=20).cuda()
        with torch.no_grad():
            for n, v in dict(alpha_w1=0.05, alpha_w2=0.02, alpha_w3=-0.01, beta_w1=0.04, beta_w2=0.015, beta_w3=0.01).items():
                getattr(pl, n).fill_(v)
        x = torch.randn(9, 1, 20, 20, generator=g).cuda()
        _repeat(_fwd_bwd(pl, x, torch.randn(9, 1, 20, 20, generator=g).cuda()))


@pytest.mark.parametrize("N", [12, 28])
def test_bf16_tensors_are_deterministic_and_finite(N):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(60 + N)
    for C, mixing in ((4, True), (6, False), (32, True)):
        layer = quiet(P.EnhancedDiffusionLayer, N, C, dt=0.05, num_steps=2, channel_mixing_enabled=mixing).cuda()
        _perturb(layer, g, 0.3)
        if not mixing:
            layer.channel_mixing.requires_grad_(False)
        u = torch.randn(19, C, N, N, generator=g).bfloat16().cuda()
        gy = torch.randn(19, C, N, N, generator=g).bfloat16().cuda()
        _repeat(_fwd_bwd(layer, u, gy), reps=3)
