"""GPU parity, gate 3: the BASELINE.json configurations as LAYERS against the CPU oracle — the composite paths
the per-kernel tests do not reach (MFMA channel operators with gradient accumulation across time steps, bf16
states, checkpoints together with smoothing and a wide coupling, the skip blend), at batch sizes the oracle
finishes in seconds:

    cfg2 secondary   cifar10.EnhancedDiffusionLayer(32, 64, num_steps=10) WITH channel mixing   fp32   1e-5
    cfg3 at 32 ch    SVHN.DiffusionLayer(28, 32, dt=0.3, num_steps=4), fashion-like coefficients  fp32   1e-5
    cfg4             SVHN.DiffusionLayer(32, 128, num_steps=20), bf16 tensors                     bf16   3e-2 / 1e-2

plus the mixed layer call against its per-step composition at every channel count that takes an MFMA path
(32, 64, 96, 128; fp32 and bf16), a second backward through a retained graph, and the checkpoint plan after a
parameter jump on a layer that has already been called."""
import contextlib
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


def _layer_vs_oracle(layer, spec, u, gy, tol, dtype=torch.float32, state_cast=None):
    params = {k: v.detach().clone() for k, v in layer.named_parameters()}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec, state_cast), u, params, gy)
    dl = layer.cuda()
    ud = u.to(dtype).cuda().requires_grad_(True)
    y = dl(ud)
    assert y.dtype == dtype and y.shape == u.shape
    y.backward(gy.to(dtype).cuda())
    torch.cuda.synchronize()
    errs = {"y": G.rel_err(y.detach().float().cpu(), y_ref), "gu": G.rel_err(ud.grad.float().cpu(), gu_ref)}
    for n, p in dl.named_parameters():
        assert p.grad is not None, n
        errs["g_" + n] = G.rel_err(p.grad.float().cpu().reshape(gp_ref[n].shape), gp_ref[n])
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, (bad, errs)
    return errs


def test_cfg2_with_channel_mixing_vs_oracle():
    """BASELINE configs[1] as the reference defines it (mixing before every step, cifar10.py:91): C = 64 takes the
    fp32-MFMA apply kernel and the fused mixing backward with partial sums accumulated over the ten steps."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(2064)
    B, C, N, steps = 4, 64, 32, 10
    layer = quiet(P.EnhancedDiffusionLayer, N, C, num_steps=steps)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    _layer_vs_oracle(layer, O.cifar10_spec(N, C, num_steps=steps), u, gy, TOL)


def test_cfg3_32_channels_vs_oracle():
    """SURVEY §8d cfg3 at 32 channels: fashion semantics (dt = 0.3, four Strang steps, smoothed coefficients) on
    the SVHN layer; coefficients of fashion size (1.8 * 0.3 = 0.54) so that the backward needs checkpoints,
    a live coupling matrix (C = 32: one-wave fused MFMA kernels) and the skip blend."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(2832)
    B, C, N = 4, 32, 28
    layer = P.SvhnDiffusionLayer(N, C, dt=0.3, dx=1.0, num_steps=4)
    with torch.no_grad():
        layer.alpha_base.copy_(1.8 * (1 + 0.1 * torch.randn(C, N, N, generator=g)))
        layer.beta_base.copy_(1.8 * (1 + 0.1 * torch.randn(C, N, N, generator=g)))
        layer.alpha_time_coeff.copy_(0.3 * torch.randn(C, N, N, generator=g))
        layer.beta_time_coeff.copy_(0.3 * torch.randn(C, N, N, generator=g))
        layer.channel_coupling.copy_(torch.eye(C) + 0.05 * torch.randn(C, C, generator=g))
        layer.skip_weight.fill_(0.3)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    _layer_vs_oracle(layer, O.svhn_spec(N, C, dt=0.3, dx=1.0, num_steps=4), u, gy, TOL)


def test_cfg4_bf16_vs_fp32_oracle():
    """BASELINE configs[3]: SVHN.DiffusionLayer(32, 128, num_steps=20) on bf16 tensors (fp32 arithmetic inside,
    SURVEY D6): 60 sweeps, 20 couplings on the bf16 MFMA with their gradient accumulated over the steps, bf16
    states between the launches, skip blend.  Two comparisons with the fp32 oracle fed the bf16-rounded input:
    (1) the oracle as it is: the layer rounds its state to bf16 41 times on the way (after each of the 20 sweep
        launches, each of the 20 couplings and the blend), each rounding up to 2^-9 of a value, so the max-norm
        error is held to 3e-2 (measured 2.3e-2 on the input gradient);
    (2) the oracle rounding its state to bf16 at those same points (``state_cast``), which removes that
        accumulation from the comparison: 1e-2, i.e. an error in the arithmetic would show.
    The coupling is I + 0.05 randn / sqrt(C/8): the default 0.01 I kills the signal and an
    unscaled 0.05 randn at C = 128 grows it by 1.6x per step."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(4128)
    B, C, N, steps = 2, 128, 32, 20
    layer = P.SvhnDiffusionLayer(N, C, num_steps=steps)
    with torch.no_grad():
        layer.alpha_base.mul_(1 + 0.2 * torch.randn(C, N, N, generator=g))
        layer.beta_base.mul_(1 + 0.2 * torch.randn(C, N, N, generator=g))
        layer.alpha_time_coeff.copy_(0.5 * torch.randn(C, N, N, generator=g))
        layer.beta_time_coeff.copy_(0.5 * torch.randn(C, N, N, generator=g))
        layer.channel_coupling.copy_(torch.eye(C) + 0.05 / 4.0 * torch.randn(C, C, generator=g))
        layer.skip_weight.fill_(0.1)
    u = torch.randn(B, C, N, N, generator=g).bfloat16().float()
    gy = torch.randn(B, C, N, N, generator=g).bfloat16().float()
    import copy
    _layer_vs_oracle(copy.deepcopy(layer), O.svhn_spec(N, C, num_steps=steps), u, gy, 3e-2, dtype=torch.bfloat16)
    _layer_vs_oracle(layer, O.svhn_spec(N, C, num_steps=steps), u, gy, 1e-2, dtype=torch.bfloat16,
                     state_cast=lambda t: t.bfloat16().float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [32, 64, 96, 128])
@pytest.mark.parametrize("mode", ["pre", "post"])
def test_mixed_call_mfma_widths_equal_per_step_composition(mode, C, dtype):
    """The mixed layer call at the channel counts that take an MFMA mixing path (fused backward with partial sums
    accumulated over K = 3 steps and reduced once; C = 128 reads its M^T fragments from a table) against the
    same steps as separate autograd nodes (single-call mixing kernels, covered by test_channel_mix_vs_fp64)."""
    import cnn_with_pde_amd as P
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(1000 + C)
    B, N, K = 3, 32, 3
    steps = P.adi_schedule(0.05, 1.0, 1.0, K, "strang")
    mk = lambda: [(1 + 0.2 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True) for _ in range(2)] + \
                 [(0.5 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True) for _ in range(2)]
    p1 = mk()
    p2 = [t.detach().clone().requires_grad_(True) for t in p1]
    M1 = (torch.eye(C) + 0.1 / (C / 8) ** 0.5 * torch.randn(C, C, generator=g)).cuda().requires_grad_(True)
    M2 = M1.detach().clone().requires_grad_(True)
    u = torch.randn(B, C, N, N, generator=g).to(dtype).cuda()
    gy = torch.randn(B, C, N, N, generator=g).to(dtype).cuda()
    u1 = u.clone().requires_grad_(True)
    y1 = F_.adi_diffuse_mixed(u1, *p1, M1, steps, mode, smooth3=True, checkpoints="auto")
    y1.backward(gy)
    u2 = u.clone().requires_grad_(True)
    v = u2
    for st in steps:
        if mode == "pre":
            v = P.adi_diffuse(P.channel_mix(v, M2), *p2, st, smooth3=True, checkpoints="auto")
        else:
            v = P.channel_mix(P.adi_diffuse(v, *p2, st, smooth3=True, checkpoints="auto"), M2)
    v.backward(gy)
    f = lambda t: t.detach().float().cpu()
    # same kernels on the same inputs in the same order: values agree to rounding of the fp32 sums
    vt, gt = (2e-6, 5e-6) if dtype == torch.float32 else (1e-2, 1e-2)
    assert G.rel_err(f(y1), f(v)) <= vt
    assert G.rel_err(f(u1.grad), f(u2.grad)) <= gt
    # the matrix gradient: accumulated partial sums vs one reduction per step and a torch add
    assert G.rel_err(f(M1.grad), f(M2.grad)) <= (1e-5 if dtype == torch.float32 else 2e-3)
    for q1, q2 in zip(p1, p2):
        assert G.rel_err(f(q1.grad), f(q2.grad)) <= (1e-5 if dtype == torch.float32 else 2e-2)


def test_mixed_node_second_backward_through_retained_graph():
    """retain_graph=True: the mixed node keeps its factorisation; a second backward gives the same gradients."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(88)
    layer = quiet(P.EnhancedDiffusionLayer, 32, 3, dt=0.02, num_steps=3).cuda()
    u = torch.randn(5, 3, 32, 32, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(5, 3, 32, 32, generator=g).cuda()
    y = layer(u)
    y.backward(gy, retain_graph=True)
    first = {n: p.grad.clone() for n, p in layer.named_parameters()}
    gu1 = u.grad.clone()
    for p in layer.parameters():
        p.grad = None
    u.grad = None
    y.backward(gy)
    assert torch.equal(u.grad, gu1)
    for n, p in layer.named_parameters():
        assert torch.equal(p.grad, first[n]), n


@pytest.mark.parametrize("policy", ["auto", "lagged"])
def test_checkpoint_plan_after_parameter_jump(policy):
    """A layer that has already run at tiny coefficients (no checkpoints needed) gets the fashion_trained parameters
    through load_state_dict: the very next backward must give the reference's gradients at 1e-5.  (Without
    checkpoints these coefficients lose the parameter gradients altogether — test_large_coefficients_need_checkpoints.)"""
    import cnn_with_pde_amd as P
    g = G.Golden("fashion_trained")
    layer = P.FashionDiffusionLayer(**g.ctor)
    layer.checkpoint_policy = policy
    with torch.no_grad():
        layer.alpha_base.fill_(1e-3)
        layer.beta_base.fill_(1e-3)
    layer = layer.cuda()
    for _ in range(2):                                   # two calls: the lagged cache is warm
        w = torch.randn(3, 1, 28, 28, device="cuda", requires_grad=True)
        layer(w).sum().backward()
    for p in layer.parameters():
        p.grad = None
    layer.load_state_dict({k: v.float() for k, v in g.params.items()})
    u = g.u.float().cuda().requires_grad_(True)
    y = layer(u)
    y.backward(g.gy.float().cuda())
    errs = {"y": G.rel_err(y.detach().cpu(), g.y), "gu": G.rel_err(u.grad.cpu(), g.gu)}
    for n, p in layer.named_parameters():
        errs["g_" + n] = G.rel_err(p.grad.cpu(), g.grads[n])
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (policy, bad, errs)
    # ... and after a change of dt on the same object (the attribute is documented as changeable)
    if policy == "lagged":
        layer.dt = 0.25
        assert "_kmax_cache" in layer.__dict__
        u2 = g.u.float().cuda().requires_grad_(True)
        layer(u2).backward(g.gy.float().cuda())
        spec = O.fashion_spec(28, 0.25, 1.0, 4)
        params = {k: v.detach().cpu() for k, v in layer.named_parameters()}
        _, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), g.u.float(), params, g.gy.float())
        assert G.rel_err(u2.grad.cpu(), gu_ref) <= TOL
