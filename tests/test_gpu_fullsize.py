"""Full BASELINE sizes of the configurations other than cfg2 (which tests/test_gpu_parity.py and test_gpu_wide.py hold):
size-independent properties that need no oracle — the layers are linear in u, so linearity, the adjoint identity
<gy, J u> = <J^T gy, u>, a slice of the batch run alone, and parameter gradients of two parts of the batch adding up to
those of the whole must hold at any size.  (At sizes the oracle finishes in seconds the same layers are compared with it
in test_gpu_configs.py / test_gpu_parity.py.)"""
import contextlib
import io

import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _properties(layer, shape, dtype, lin_tol, adj_tol, add_tol, cut, linear=True):
    g = torch.Generator().manual_seed(31)
    B = shape[0]
    u = torch.randn(*shape, generator=g).to(dtype).cuda()
    v = torch.randn(*shape, generator=g).to(dtype).cuda()
    gy = torch.randn(*shape, generator=g).to(dtype).cuda()
    with torch.no_grad():
        yu = layer(u)
        if linear:
            yv = layer(v)
            lhs = layer((2.0 * u.float() - 0.5 * v.float()).to(dtype))
            rhs = 2.0 * yu.float() - 0.5 * yv.float()
            assert float((lhs.float() - rhs).abs().max() / rhs.abs().max()) <= lin_tol
            del yv, lhs, rhs
        sl = slice(B // 3, B // 3 + 24)
        assert torch.equal(layer(u[sl].contiguous()), yu[sl])          # a sample does not depend on its batch
    del yu, v
    for p in layer.parameters():
        p.grad = None
    ud = u.clone().requires_grad_(True)
    y = layer(ud)
    y.backward(gy)
    if linear:
        a = float((y.detach().double() * gy.double()).sum())
        b = float((ud.grad.double() * u.double()).sum())
        assert abs(a - b) <= adj_tol * float(y.detach().double().norm() * gy.double().norm())
    full = {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}
    assert all(bool(torch.isfinite(t).all()) for t in full.values()) and bool(torch.isfinite(ud.grad).all())
    del y, ud
    acc = {n: torch.zeros_like(t) for n, t in full.items()}
    for part in (slice(0, cut), slice(cut, B)):
        for p in layer.parameters():
            p.grad = None
        uh = u[part].clone().requires_grad_(True)
        layer(uh).backward(gy[part])
        for n, p in layer.named_parameters():
            if p.grad is not None:
                acc[n] += p.grad
    errs = {n: G.rel_err(acc[n].float().cpu(), full[n].float().cpu()) for n in full}
    bad = {n: e for n, e in errs.items() if not e <= add_tol}
    assert not bad, (bad, errs)


def test_cfg3_fashion_single_channel_full_batch():
    """BASELINE configs[2] literally: fashion_mnist.DiffusionLayer() on (4096, 1, 28, 28) — coefficients 0.27/0.54, so the
    backward runs with checkpoints."""
    import cnn_with_pde_amd as P
    layer = quiet(P.FashionDiffusionLayer).cuda()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        layer.alpha_base.mul_((1 + 0.1 * torch.randn(layer.alpha_base.shape, generator=g)).cuda())
        layer.beta_time_coeff.copy_((0.2 * torch.randn(layer.beta_time_coeff.shape, generator=g)).cuda())
    _properties(layer, (4096, 1, 28, 28), torch.float32, 5e-6, 1e-5, 1e-5, 1500)


def test_cfg3_at_32_channels_full_batch():
    """SURVEY §8d cfg3 at 32 channels: SVHN.DiffusionLayer(28, 32, dt=0.3, num_steps=4) semantics, coupling every step and
    the skip blend, (512, 32, 28, 28)."""
    import cnn_with_pde_amd as P
    layer = quiet(P.SvhnDiffusionLayer, 28, 32, dt=0.3, num_steps=4)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        layer.alpha_base.fill_(1.8); layer.beta_base.fill_(1.8)
        layer.channel_coupling.copy_(torch.eye(32) + 0.05 * torch.randn(32, 32, generator=g))
        layer.skip_weight.fill_(0.3)
    _properties(layer.cuda(), (512, 32, 28, 28), torch.float32, 5e-6, 1e-5, 1e-5, 200)


def test_cfg4_svhn_128_channels_bf16_full_batch():
    """BASELINE configs[3]: SVHN.DiffusionLayer(32, 128, num_steps=20) on bf16 tensors, (512, 128, 32, 32): 60 sweeps,
    20 couplings on the bf16 matrix cores, skip blend.  bf16 tensors round the state 41 times: linearity and the adjoint
    identity hold to bf16 accuracy; additivity of the fp32 parameter gradients to 2e-3 (their inputs are bf16-rounded
    per part identically, only the summation order differs)."""
    import cnn_with_pde_amd as P
    layer = quiet(P.SvhnDiffusionLayer, 32, 128, num_steps=20)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        layer.channel_coupling.copy_(torch.eye(128) + 0.01 * torch.randn(128, 128, generator=g))
    _properties(layer.cuda(), (512, 128, 32, 32), torch.bfloat16, 3e-2, 2e-2, 2e-3, 200)


def test_cfg5_tiny_imagenet_explicit_full_per_gpu_batch():
    """BASELINE configs[4] at its per-GPU batch: tiny_imagenet.ImprovedDiffusionLayer(64, 64) on (256, 64, 64, 64).  The
    layer is affine-free in u (u + 0.1 (s u + a dt Lap(s u) - u)): linear."""
    import cnn_with_pde_amd as P
    layer = P.ImprovedDiffusionLayer(64, 64)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        layer.alpha_base.copy_(0.2 * torch.rand(64, generator=g))
        layer.channel_scaling.copy_(1 + 0.2 * torch.randn(64, generator=g))
    _properties(layer.cuda(), (256, 64, 64, 64), torch.float32, 5e-6, 1e-5, 1e-5, 100)
