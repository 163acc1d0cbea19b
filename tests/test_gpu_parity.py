"""GPU parity, gate 2: the HIP path (through the C ABI) against the CPU oracle on seeded inputs at
sizes the oracle finishes in seconds, plus edge cases (ragged / single-sample batches, odd channel
counts, explicit / automatic / no checkpoints, bf16 I/O, schedules that are neither Strang nor Lie)
and, at BASELINE.json's full size, size-independent properties (linearity, constants, the adjoint
identity <gy, J u> = <J^T gy, u>, additivity of parameter gradients over the batch).

Tolerances: fp32 1e-5 relative (max-norm) as north_star states; bf16 I/O 2e-2 (bf16 has 8 bits)."""
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


def _perturb(layer, g, rel=0.15, slope=0.0):
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + rel * torch.randn(p.shape, generator=g))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(slope * torch.randn(p.shape, generator=g))


def _compare(layer, spec_fn, u, gy, tol=TOL, dtype=torch.float32):
    params = {k: v.detach().clone() for k, v in layer.named_parameters()}
    y_ref, gu_ref, gp_ref = O.value_and_grads(spec_fn, u, params, gy)
    dl = layer.cuda()
    ud = u.to(dtype).cuda().requires_grad_(True)
    y = dl(ud)
    assert y.dtype == dtype and y.shape == u.shape
    y.backward(gy.to(dtype).cuda())
    torch.cuda.synchronize()
    errs = {"y": G.rel_err(y.detach().float().cpu(), y_ref), "gu": G.rel_err(ud.grad.float().cpu(), gu_ref)}
    for n, p in dl.named_parameters():
        if gp_ref.get(n) is not None and p.grad is not None:
            errs["g_" + n] = G.rel_err(p.grad.cpu(), gp_ref[n])
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, (bad, errs)
    return errs


@pytest.mark.parametrize("B,C,N,steps,dt,slope", [
    (13, 3, 32, 4, 0.01, 0.5),       # ragged batch: not a multiple of the planes per workgroup iteration
    (1, 1, 32, 2, 0.02, 0.0),        # single sample, single channel
    (40, 5, 28, 3, 0.02, 1.0),       # odd channel count, N = 28
    (70, 2, 16, 5, 0.05, 0.0),       # several chunks per workgroup
])
def test_cifar10_semantics_vs_oracle(B, C, N, steps, dt, slope):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(100 + B)
    layer = quiet(P.EnhancedDiffusionLayer, N, C, dt=dt, num_steps=steps)
    _perturb(layer, g, 0.15, slope)
    spec = O.cifar10_spec(N, C, dt=dt, num_steps=steps)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    _compare(layer, lambda a, p: O.adi_forward(a, p, spec), u, gy)


@pytest.mark.parametrize("which", ["mnist", "fashion", "svhn", "cifar2"])
def test_other_variants_vs_oracle(which):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(7)
    if which == "mnist":
        layer, spec, C, N = quiet(P.MnistDiffusionLayer, 28, 0.004, 0.8, 1.3, 6), O.mnist_spec(28, 0.004, 0.8, 1.3, 6), 1, 28
    elif which == "fashion":
        layer, spec, C, N = P.FashionDiffusionLayer(28, 0.3, 1.0, 4), O.fashion_spec(28, 0.3, 1.0, 4), 1, 28
    elif which == "svhn":
        layer, spec, C, N = P.SvhnDiffusionLayer(32, 4, 0.05, 1.0, 3), O.svhn_spec(32, 4, 0.05, 1.0, 3), 4, 32
        with torch.no_grad():
            layer.channel_coupling.copy_(torch.eye(4) + 0.05 * torch.randn(4, 4, generator=g))
            layer.skip_weight.fill_(0.2)
    else:
        layer, spec, C, N = quiet(P.LearnableDiffusionLayer, 32, 3, 0.02, 1.0, 1.5, 5), O.cifar2_spec(32, 3, 0.02, 1.0, 1.5, 5), 3, 32
    _perturb(layer, g, 0.2, 0.3)
    u = torch.randn(19, C, N, N, generator=g)
    gy = torch.randn(19, C, N, N, generator=g)
    _compare(layer, lambda a, p: O.adi_forward(a, p, spec), u, gy)


@pytest.mark.parametrize("N", [8, 12, 16, 20, 24, 28, 32])
def test_every_line_length_vs_oracle(N):
    """One kernel instantiation per supported N (multiples of 4 up to 32): Lie and Strang splits, smoothing,
    idle lanes of the 32-line wave (N < 32) and a batch that is not a multiple of the planes per iteration."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(200 + N)
    lie = quiet(P.LearnableDiffusionLayer, N, 2, 0.03, 1.0, 1.2, 3)
    _perturb(lie, g, 0.2, 0.4)
    u = torch.randn(21, 2, N, N, generator=g)
    gy = torch.randn(21, 2, N, N, generator=g)
    _compare(lie, lambda a, p: O.adi_forward(a, p, O.cifar2_spec(N, 2, 0.03, 1.0, 1.2, 3)), u, gy)
    strang = quiet(P.MnistDiffusionLayer, N, 0.01, 1.0, 1.0, 3)           # smooth3, C = 1
    _perturb(strang, g, 0.2, 0.4)
    u = torch.randn(5, 1, N, N, generator=g)
    gy = torch.randn(5, 1, N, N, generator=g)
    _compare(strang, lambda a, p: O.adi_forward(a, p, O.mnist_spec(N, 0.01, 1.0, 1.0, 3)), u, gy)


def test_schedules_longer_than_one_launch():
    """num_steps = 40 Strang steps = 120 sweeps > PDE_MAX_SWEEPS: the layers chain two launch sequences."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(404)
    layer = quiet(P.EnhancedDiffusionLayer, 32, 2, dt=0.004, num_steps=40, channel_mixing_enabled=False)
    _perturb(layer, g, 0.2, 0.3)
    with torch.no_grad():
        layer.channel_mixing.copy_(torch.eye(2))
    u = torch.randn(3, 2, 32, 32, generator=g)
    gy = torch.randn(3, 2, 32, 32, generator=g)
    _compare(layer, lambda a, p: O.adi_forward(a, p, O.cifar10_spec(32, 2, dt=0.004, num_steps=40)), u, gy, tol=2e-5)
    mixed = quiet(P.EnhancedDiffusionLayer, 32, 2, dt=0.004, num_steps=40)
    _perturb(mixed, g, 0.2, 0.3)
    _compare(mixed, lambda a, p: O.adi_forward(a, p, O.cifar10_spec(32, 2, dt=0.004, num_steps=40)), u, gy, tol=2e-5)


def test_unsupported_sizes_fail_loudly():
    """N beyond what the any-size path holds in LDS is an error from the C ABI, never a silent other path
    (line lengths 2..128 without fused kernels: tests/test_gpu_anysize.py)."""
    import cnn_with_pde_amd as P
    from cnn_with_pde_amd._lib import PdeError
    for N in (1, 129, 200):
        layer = quiet(P.MnistDiffusionLayer, N).cuda()
        with pytest.raises(PdeError):
            layer(torch.zeros(2, 1, N, N, device="cuda"))


def test_bf16_io_vs_fp32_oracle():
    """bf16 tensors in and out, fp32 arithmetic inside: compared with the fp32 oracle fed
    the bf16-rounded input, at bf16 resolution."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(21)
    layer = quiet(P.EnhancedDiffusionLayer, 32, 6, dt=0.02, num_steps=3, channel_mixing_enabled=False)
    _perturb(layer, g, 0.1, 0.2)
    with torch.no_grad():
        layer.channel_mixing.copy_(torch.eye(6))
    spec = O.cifar10_spec(32, 6, dt=0.02, num_steps=3)
    u = torch.randn(9, 6, 32, 32, generator=g).bfloat16().float()
    gy = torch.randn(9, 6, 32, 32, generator=g).bfloat16().float()
    _compare(layer, lambda a, p: O.adi_forward(a, p, spec), u, gy, tol=2e-2, dtype=torch.bfloat16)


def test_checkpoint_modes_agree_and_match_oracle():
    """No checkpoints, every state checkpointed, and the automatic plan give the same gradients."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(31)
    N, C, B, steps, dt = 32, 2, 10, 3, 0.01
    spec = O.cifar10_spec(N, C, dt=dt, num_steps=steps)
    params = O.adi_init_params(spec, "cifar10", gen=g)
    params["channel_mixing"] = torch.eye(C)
    for k in ("alpha_base", "beta_base"):
        params[k] = params[k] * (1 + 0.2 * torch.randn(params[k].shape, generator=g))
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    _, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    sweeps = [s for st in P.adi_schedule(dt, 1.0, 1.0, steps) for s in st]
    S = len(sweeps)
    names = ["alpha_base", "beta_base", "alpha_time_coeff", "beta_time_coeff"]
    for ck in (0, (1 << (S - 1)) - 1, 0b10010, "auto"):
        ud = u.cuda().requires_grad_(True)
        ps = [params[k].cuda().requires_grad_(True) for k in names]
        y = P.adi_diffuse(ud, *ps, sweeps, smooth3=False, clamp_max=10.0, checkpoints=ck)
        y.backward(gy.cuda())
        assert G.rel_err(ud.grad.cpu(), gu_ref) <= TOL, ck
        for k, p in zip(names, ps):
            assert G.rel_err(p.grad.cpu(), gp_ref[k]) <= TOL, (ck, k)


def test_large_coefficients_need_checkpoints():
    """fashion-like coefficients: rebuilding every state from the output loses the gradient, the
    planned checkpoints keep it (this is why the plan exists)."""
    import cnn_with_pde_amd as P
    g = G.Golden("fashion_default")
    sweeps = [s for st in P.adi_schedule(0.3, 1.0, 1.0, 4) for s in st]
    names = ["alpha_base", "beta_base", "alpha_time_coeff", "beta_time_coeff"]
    errs = {}
    for ck in (0, "auto"):
        ud = g.u.cuda().requires_grad_(True)
        ps = [g.params[k].cuda().requires_grad_(True) for k in names]
        P.adi_diffuse(ud, *ps, sweeps, smooth3=True, checkpoints=ck).backward(g.gy.cuda())
        errs[ck] = max(G.rel_err(p.grad.cpu(), g.grads[k]) for k, p in zip(names, ps))
    assert errs["auto"] <= TOL and errs[0] > errs["auto"], errs


def test_generic_schedule_equals_composition():
    """A fused run whose axes follow neither split (x,y,x,x,y) takes the table-driven kernels; it must
    equal the Strang run followed by the Lie run, values and gradients."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(41)
    C, N, B = 3, 32, 11
    mk = lambda: [(1 + 0.2 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True) for _ in range(2)] + \
                 [(0.5 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True) for _ in range(2)]
    p1 = mk()
    p2 = [t.detach().clone().requires_grad_(True) for t in p1]
    a = P.adi_schedule(0.02, 1.0, 1.0, 1, "strang")[0]
    # (x, y) with the same per-axis increments as the Strang step: one launch may not mix different
    # delta/h2 on one axis (pdecnn.h), which no reference variant does either
    b = [P.Sweep(0, 0.01, 1.0, 0.02), P.Sweep(1, 0.02, 1.0, 0.03)]
    u = torch.randn(B, C, N, N, generator=g).cuda()
    gy = torch.randn(B, C, N, N, generator=g).cuda()
    u1 = u.clone().requires_grad_(True)
    y1 = P.adi_diffuse(u1, *p1, a + b, clamp_max=10.0)
    y1.backward(gy)
    u2 = u.clone().requires_grad_(True)
    y2 = P.adi_diffuse(P.adi_diffuse(u2, *p2, a, clamp_max=10.0), *p2, b, clamp_max=10.0)
    y2.backward(gy)
    assert G.rel_err(y1.detach().cpu(), y2.detach().cpu()) <= 2e-6
    assert G.rel_err(u1.grad.cpu(), u2.grad.cpu()) <= 2e-6
    for q1, q2 in zip(p1, p2):
        assert G.rel_err(q1.grad.cpu(), q2.grad.cpu()) <= TOL


@pytest.mark.parametrize("mode,split,ck", [("pre", "strang", 0), ("pre", "lie", 0b1), ("post", "strang", 0b11), ("post", "strang", "auto")])
def test_mixed_layer_call_equals_per_step_composition(mode, split, ck):
    """adi_diffuse_mixed (one factorisation, per-step launches, gradients accumulated on the device) against
    the same steps written as separate autograd nodes (adi_diffuse + channel_mix per step)."""
    import cnn_with_pde_amd as P
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(77)
    B, C, N, K = (9, 5, 32, 4) if ck != 0b1 else (1, 2, 28, 1)      # also: a single step, a single sample
    steps = P.adi_schedule(0.05, 1.0, 1.0, K, split)
    mk = lambda: [(1 + 0.2 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True) for _ in range(2)] + \
                 [(0.5 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True) for _ in range(2)]
    p1 = mk()
    p2 = [t.detach().clone().requires_grad_(True) for t in p1]
    M1 = (torch.eye(C) + 0.1 * torch.randn(C, C, generator=g)).cuda().requires_grad_(True)
    M2 = M1.detach().clone().requires_grad_(True)
    u = torch.randn(B, C, N, N, generator=g).cuda()
    gy = torch.randn(B, C, N, N, generator=g).cuda()
    u1 = u.clone().requires_grad_(True)
    y1 = F_.adi_diffuse_mixed(u1, *p1, M1, steps, mode, smooth3=True, checkpoints=ck)
    y1.backward(gy)
    u2 = u.clone().requires_grad_(True)
    v = u2
    for st in steps:
        if mode == "pre":
            v = P.adi_diffuse(P.channel_mix(v, M2), *p2, st, smooth3=True, checkpoints="auto")
        else:
            v = P.channel_mix(P.adi_diffuse(v, *p2, st, smooth3=True, checkpoints="auto"), M2)
    v.backward(gy)
    assert G.rel_err(y1.detach().cpu(), v.detach().cpu()) <= 2e-6
    assert G.rel_err(u1.grad.cpu(), u2.grad.cpu()) <= 5e-6
    assert G.rel_err(M1.grad.cpu(), M2.grad.cpu()) <= 1e-5
    for q1, q2 in zip(p1, p2):
        assert G.rel_err(q1.grad.cpu(), q2.grad.cpu()) <= 1e-5


@pytest.mark.parametrize("C,HW", [(3, 1024), (64, 1024), (7, 49), (32, 784), (128, 256), (96, 64)])
def test_channel_mix_vs_fp64(C, HW):
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(C)
    B = 5
    u = torch.randn(B, C, HW, generator=g)
    M = torch.eye(C) + 0.1 * torch.randn(C, C, generator=g)
    go = torch.randn(B, C, HW, generator=g)
    ud, Md = u.cuda().requires_grad_(True), M.cuda().requires_grad_(True)
    out = P.channel_mix(ud.view(B, C, HW, 1), Md)
    out.backward(go.view(B, C, HW, 1).cuda())
    u64, M64 = u.double().requires_grad_(True), M.double().requires_grad_(True)
    ref = torch.matmul(M64, u64)
    ref.backward(go.double())
    assert G.rel_err(out.detach().cpu().view(B, C, HW), ref.detach()) <= TOL
    assert G.rel_err(ud.grad.cpu(), u64.grad) <= TOL
    assert G.rel_err(Md.grad.cpu(), M64.grad) <= TOL


@pytest.mark.parametrize("C,B,HW,scale", [(64, 5, 1024, 0.1), (64, 64, 1024, 0.3), (64, 7, 64, 1.0), (64, 1, 4096, 0.05),
                                          (32, 5, 784, 0.1), (32, 64, 1024, 0.3), (32, 3, 16, 1.0), (96, 5, 1024, 0.1),
                                          (96, 3, 196, 0.3), (64, 5, 784, 0.1), (96, 2, 4, 0.1)])
def test_channel_mix_three_piece_products(C, B, HW, scale):
    """fp32 tensors at C = 32, 64, 96 run the operator's backward on the bf16 matrix cores with every operand as three bf16
    pieces (pde_mix_bf16.hip: mix_bwd_split_kernel; planes of any multiple of four pixels — 28 x 28 = 784 leaves a ragged
    last tile).  That must be fp32 arithmetic to the last bits, not bf16 arithmetic: against fp64, with six decades of
    dynamic range across the channels of the incoming gradient, tighter than the 1e-5 of the other paths (measured 6-9e-8
    and 2-4e-7; torch's own fp32 matmul is at 7e-8 and 3-7e-7)."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(1000 + B)
    u = torch.randn(B, C, HW, generator=g)
    M = torch.eye(C) + scale * torch.randn(C, C, generator=g)
    go = torch.randn(B, C, HW, generator=g) * torch.logspace(-3, 3, C).view(1, C, 1)
    ud, Md = u.cuda().requires_grad_(True), M.cuda().requires_grad_(True)
    out = P.channel_mix(ud.view(B, C, HW, 1), Md)
    out.backward(go.view(B, C, HW, 1).cuda())
    u64, M64 = u.double().requires_grad_(True), M.double().requires_grad_(True)
    ref = torch.matmul(M64, u64)
    ref.backward(go.double())
    assert G.rel_err(ud.grad.cpu().double(), u64.grad) <= 5e-7
    assert G.rel_err(Md.grad.cpu().double(), M64.grad) <= 2e-6
    # per-channel accuracy too: a small channel of gu must not drown in the rounding of a large one
    per = ((ud.grad.cpu().double() - u64.grad).abs().amax(dim=(0, 2)) / u64.grad.abs().amax(dim=(0, 2))).max()
    assert float(per) <= 2e-6, float(per)


@pytest.mark.parametrize("C", [64, 128, 32])
def test_channel_mix_bf16_io(C):
    """bf16 tensors through the MFMA mixing kernels (C = 64/128: fused backward), fp32 arithmetic inside."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(300 + C)
    B, HW = 3, 1024
    u = torch.randn(B, C, HW, generator=g).bfloat16()
    M = torch.eye(C) + 0.05 * torch.randn(C, C, generator=g)
    go = torch.randn(B, C, HW, generator=g).bfloat16()
    ud, Md = u.cuda().view(B, C, 32, 32).requires_grad_(True), M.cuda().requires_grad_(True)
    out = P.channel_mix(ud, Md)
    assert out.dtype == torch.bfloat16
    out.backward(go.cuda().view(B, C, 32, 32))
    u64, M64 = u.double().requires_grad_(True), M.double().requires_grad_(True)
    ref = torch.matmul(M64, u64)
    ref.backward(go.double())
    assert G.rel_err(out.detach().float().cpu().view(B, C, HW), ref.detach()) <= 6e-3      # one bf16 rounding of the result
    assert G.rel_err(ud.grad.float().cpu().view(B, C, HW), u64.grad) <= 6e-3
    # C = 64 / 128 run on the bf16 MFMA: products of bf16 inputs are exact there, the sum is fp32
    assert G.rel_err(Md.grad.cpu(), M64.grad) <= (2e-5 if C in (64, 128) else 1e-4)


def test_explicit_layers_vs_oracle():
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(51)
    layer = P.ImprovedDiffusionLayer(32, 5, dt=0.3, num_steps=2)
    with torch.no_grad():
        layer.alpha_base.copy_(torch.tensor([0.05, 0.2, -0.1, 0.1, 0.149]))
        layer.channel_scaling.copy_(1 + 0.2 * torch.randn(5, generator=g))
    u = torch.randn(7, 5, 32, 32, generator=g)
    gy = torch.randn(7, 5, 32, 32, generator=g)
    _compare(layer, lambda a, p: O.tiny_forward(a, p, dt=0.3, num_steps=2), u, gy)
    # emotion layer: the kernel's own gradients are the coefficient VECTORS' (rows / columns) ...
    H, W, nt = 40, 36, 4
    u = torch.randn(6, H, W, generator=g)
    gy = torch.randn(6, H, W, generator=g)
    A = 0.04 + 0.02 * torch.randn(H, generator=g)
    Bc = 0.05 + 0.02 * torch.randn(W, generator=g)
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.jacobi_forward(a, p["A"], p["B"], nt), u,
                                              {"A": A, "B": Bc}, gy)
    ud, Ad, Bd = u.cuda().requires_grad_(True), A.cuda().requires_grad_(True), Bc.cuda().requires_grad_(True)
    y = P.jacobi_diffuse(ud, Ad, Bd, nt)
    y.backward(gy.cuda())
    assert G.rel_err(y.detach().cpu(), y_ref) <= TOL and G.rel_err(ud.grad.cpu(), gu_ref) <= TOL
    assert G.rel_err(Ad.grad.cpu(), gp_ref["A"]) <= TOL and G.rel_err(Bd.grad.cpu(), gp_ref["B"]) <= TOL
    # ... and the module's six scalars are weighted sums of those with sin/cos weights of both signs:
    # heavy cancellation, so they are held to 2e-4 (the vectors above carry the 1e-5 bar)
    pl = P.PDELayer(Nx=40, Ny=40, T=0.004)
    with torch.no_grad():
        for n, v in dict(alpha_w1=0.04, alpha_w2=0.01, alpha_w3=0.02, beta_w1=0.05, beta_w2=-0.01, beta_w3=0.01).items():
            getattr(pl, n).fill_(v)
    u = torch.randn(6, 1, 40, 40, generator=g)
    gy = torch.randn(6, 1, 40, 40, generator=g)
    _compare(pl, lambda a, p: O.emotion_forward(a, p, Nx=40, Ny=40, T=0.004), u, gy, tol=2e-4)


@pytest.mark.parametrize("size,steps,dtype", [(64, 1, torch.float32), (64, 3, torch.float32), (32, 4, torch.float32),
                                              (16, 2, torch.float32), (24, 3, torch.float32), (40, 1, torch.float32),
                                              (64, 2, torch.bfloat16), (24, 2, torch.bfloat16)])
def test_explicit5_plane_sizes_and_steps_vs_oracle(size, steps, dtype):
    """tiny_imagenet layer: the wave-per-plane kernels (64, 32, 16: plane in registers, all steps in one launch) and
    the generic kernel (other sizes, one launch per step), ragged plane counts (B*C not a multiple of the 4 planes
    of a workgroup), clamped and unclamped alpha."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(500 + size + steps)
    C, B = 3, 7
    layer = P.ImprovedDiffusionLayer(size, C, dt=0.5, num_steps=steps)
    with torch.no_grad():
        layer.alpha_base.copy_(torch.tensor([0.05, 0.3, 0.12]))
        layer.channel_scaling.copy_(1 + 0.2 * torch.randn(C, generator=g))
    u = torch.randn(B, C, size, size, generator=g)
    gy = torch.randn(B, C, size, size, generator=g)
    if dtype == torch.bfloat16:
        u, gy = u.bfloat16().float(), gy.bfloat16().float()
    _compare(layer, lambda a, p: O.tiny_forward(a, p, dt=0.5, num_steps=steps), u, gy,
             tol=TOL if dtype == torch.float32 else 2e-2, dtype=dtype)


@pytest.mark.parametrize("amp_dtype", [torch.float16, torch.bfloat16])
def test_inside_autocast(amp_dtype):
    """cifar10.py:458-467 trains under autocast: a conv in front hands the layer a half tensor.  fp16 is
    computed in fp32 (as the reference's elementwise ops are), bf16 is taken as I/O type; gradients reach
    the conv and the layer's parameters either way."""
    import cnn_with_pde_amd as P
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(3, 4, 3, padding=1).cuda()
    layer = quiet(P.EnhancedDiffusionLayer, 32, 4, dt=0.01, num_steps=2).cuda()
    x = torch.randn(6, 3, 32, 32, device="cuda")
    with torch.autocast("cuda", dtype=amp_dtype):
        h = conv(x)
        assert h.dtype == amp_dtype
        y = layer(h)
        loss = y.float().square().mean()
    loss.backward()
    assert y.dtype in (torch.float32, amp_dtype) and torch.isfinite(y.float()).all()
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all() and conv.weight.grad.abs().max() > 0
    for n, p in layer.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    # against the same computation without autocast, at half precision resolution
    h32 = conv(x).detach()
    y32 = layer(h32)
    assert G.rel_err(y.detach().float().cpu(), y32.detach().cpu()) <= (2e-2 if amp_dtype == torch.bfloat16 else 5e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_skip_blend_vs_torch(dtype):
    """SVHN.py:73-74 in one pass, against the torch expression (values and all three gradients)."""
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(11)
    shape = (7, 3, 32, 32)                      # 21504 elements, plus a ragged size below
    for n_extra in (0, 5):
        u0 = torch.randn(*shape, generator=g)
        u = torch.randn(*shape, generator=g)
        gy = torch.randn(*shape, generator=g)
        if n_extra:
            u0, u, gy = (x.flatten()[:-3].clone() for x in (u0, u, gy))
        w = torch.tensor(0.9)
        a, b, ww = (x.double().requires_grad_(True) for x in (u0.to(dtype), u.to(dtype), w))
        s = torch.sigmoid(ww)
        ref = s * a + (1 - s) * b
        ref.backward(gy.to(dtype).double())
        ad, bd = u0.to(dtype).cuda().requires_grad_(True), u.to(dtype).cuda().requires_grad_(True)
        wd = w.cuda().requires_grad_(True)
        out = F_.skip_blend(ad, bd, wd)
        assert out.dtype == dtype
        out.backward(gy.to(dtype).cuda())
        tol = 1e-6 if dtype == torch.float32 else 6e-3
        assert G.rel_err(out.detach().float().cpu(), ref.detach()) <= tol
        assert G.rel_err(ad.grad.float().cpu(), a.grad) <= tol and G.rel_err(bd.grad.float().cpu(), b.grad) <= tol
        assert abs(float(wd.grad) - float(ww.grad)) <= 2e-5 * max(1.0, abs(float(ww.grad)))


def test_empty_batch_passes_through():
    """B = 0 (the last, empty shard of a ragged split): the reference's torch ops return an empty tensor
    and zero parameter gradients; so do the layers, without a launch."""
    import cnn_with_pde_amd as P
    layers = [(quiet(P.MnistDiffusionLayer, 28), (0, 1, 28, 28)),
              (quiet(P.SvhnDiffusionLayer, 32, 4, num_steps=2), (0, 4, 32, 32)),
              (quiet(P.EnhancedDiffusionLayer, 32, 3, num_steps=2), (0, 3, 32, 32)),
              (quiet(P.ImprovedDiffusionLayer, 32, 5), (0, 5, 32, 32)),
              (quiet(P.PDELayer, Nx=40, Ny=40), (0, 1, 40, 40))]
    for layer, shape in layers:
        layer = layer.cuda()
        u = torch.zeros(*shape, device="cuda", requires_grad=True)
        y = layer(u)
        assert y.shape == u.shape
        y.sum().backward()
        assert u.grad is not None and u.grad.shape == u.shape
        for n, p in layer.named_parameters():
            if p.grad is not None:
                assert float(p.grad.abs().max()) == 0.0, (type(layer).__name__, n)


# ---- full BASELINE size: properties that do not need the oracle -------------------------------
@pytest.fixture(scope="module")
def big():
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(1234)
    B, C, N = 512, 64, 32
    layer = quiet(P.EnhancedDiffusionLayer, N, C, num_steps=10, channel_mixing_enabled=False)
    _perturb(layer, g, 0.1, 0.1)
    layer = layer.cuda()
    u = torch.randn(B, C, N, N, generator=g).cuda()
    v = torch.randn(B, C, N, N, generator=g).cuda()
    gy = torch.randn(B, C, N, N, generator=g).cuda()
    return layer, u, v, gy


def test_full_size_linearity_and_constants(big):
    layer, u, v, gy = big
    with torch.no_grad():
        lhs = layer(2.0 * u - 0.5 * v)
        rhs = 2.0 * layer(u) - 0.5 * layer(v)
        assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= 5e-6
        c = torch.full_like(u, 3.0)
        out = layer(c)
        want = 3.0 / (1.0 + 1e-6) ** 30                       # row sums of every sweep matrix are 1
        assert float((out - want).abs().max()) <= TOL * 3.0          # 30 fp32 solves in a row


def test_full_size_adjoint_identity_and_additivity(big):
    layer, u, v, gy = big
    for p in layer.parameters():
        p.grad = None
    ud = u.clone().requires_grad_(True)
    y = layer(ud)
    y.backward(gy)
    # the layer is linear in u: <gy, J u> = <J^T gy, u>
    lhs = float((y.detach().double() * gy.double()).sum())
    rhs = float((ud.grad.double() * u.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), float(y.detach().double().norm() * gy.double().norm()) * 1e-2)
    full = {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}
    # parameter gradients are sums over the batch: two halves add up to the whole
    acc = {n: torch.zeros_like(gr) for n, gr in full.items()}
    for sl in (slice(0, 200), slice(200, 512)):
        for p in layer.parameters():
            p.grad = None
        uh = u[sl].clone().requires_grad_(True)
        layer(uh).backward(gy[sl])
        for n, p in layer.named_parameters():
            if p.grad is not None:
                acc[n] += p.grad
    for n in full:
        assert G.rel_err(acc[n].cpu(), full[n].cpu()) <= TOL, n


def test_mid_size_against_c_oracle():
    """BASELINE configs[1] semantics at a size the C restatement (closed-form gradients, OpenMP)
    finishes in seconds: 48 x 16 x 32 x 32, 10 Strang steps, trained-like coefficients."""
    from oracle import c_oracle as CO
    if not CO.available():
        pytest.skip("oracle/libpde_oracle_c.so not built")
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(77)
    B, C, N, steps = 48, 16, 32, 10
    layer = quiet(P.EnhancedDiffusionLayer, N, C, num_steps=steps, channel_mixing_enabled=False)
    _perturb(layer, g, 0.1, 0.1)
    spec = O.AdiSpec(N, C, 0.001, 1.0, 1.0, steps, "strang", False, 10.0, "none", False)
    params = {k: v.detach().clone() for k, v in layer.named_parameters() if k != "channel_mixing"}
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    y_ref, gu_ref, gp_ref = CO.adi_value_and_grads(u.double(), {k: v.double() for k, v in params.items()},
                                                   gy.double(), spec)
    dl = layer.cuda()
    ud = u.cuda().requires_grad_(True)
    y = dl(ud)
    y.backward(gy.cuda())
    assert G.rel_err(y.detach().cpu(), y_ref) <= TOL
    assert G.rel_err(ud.grad.cpu(), gu_ref) <= TOL
    for k in gp_ref:
        assert G.rel_err(getattr(dl, k).grad.cpu(), gp_ref[k]) <= TOL, k
