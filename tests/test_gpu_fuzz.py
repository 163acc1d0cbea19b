"""GPU parity, randomised: layer kind, line length, channel count, steps, step size, batch, coefficient scale and
time slopes drawn from a seeded generator, every case against the CPU oracle at 1e-5 (output, input gradient, every
parameter gradient).  The hand-picked cases of the other files cover the corners somebody thought of; this one walks
the combinations in between (C from the one-launch kernels through the scalar and MFMA operators to the one-launch
forward at 32, N with and without idle lanes, coefficients from no checkpoint to one per sweep).
``PDE_FUZZ_CASES`` / ``PDE_FUZZ_SEED`` widen the walk for a manual run."""
import contextlib
import io
import os
import random
import zlib

import pytest
import torch

import golden_util as G
from oracle import pde_oracle as O

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("PDE_FUZZ_CASES", "36"))
SEED = int(os.environ.get("PDE_FUZZ_SEED", "20260"))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _draw(rng):
    kind = rng.choice(["mnist", "fashion", "cifar10", "cifar10", "cifar2", "svhn", "svhn", "plain"])
    N = rng.choice([8, 12, 16, 20, 24, 28, 32])
    if kind in ("mnist", "fashion"):
        C = 1
    else:
        C = rng.choice([1, 2, 3, 4, 5, 8, 16, 32])
    if C == 32 and N < 28:
        N = rng.choice([28, 32])                 # keep the oracle's cost in seconds; 28/32 is where C = 32 has its own path
    steps = rng.randint(1, 4)
    dt = rng.choice([0.002, 0.02, 0.1, 0.4])
    dx = rng.choice([1.0, 1.5, 2.0])
    scale = rng.choice([0.3, 1.0, 3.0])          # coefficient scale: with dt = 0.4 and scale 3 every sweep is checkpointed
    slope = rng.choice([0.0, 0.3, 5.0])          # 5: clamp masks change inside the time window
    B = rng.choice([1, 2, 3, 5, 9]) if C >= 16 else rng.choice([1, 3, 7, 17, 40])
    return kind, N, C, steps, dt, dx, scale, slope, B


def _build(kind, N, C, steps, dt, dx):
    import cnn_with_pde_amd as P
    if kind == "mnist":
        return quiet(P.MnistDiffusionLayer, N, dt=dt, dx=dx, dy=dx, num_steps=steps), O.mnist_spec(N, dt, dx, dx, steps)
    if kind == "fashion":
        return P.FashionDiffusionLayer(N, dt=dt, dx=dx, num_steps=steps), O.fashion_spec(N, dt, dx, steps)
    if kind == "svhn":
        return P.SvhnDiffusionLayer(N, C, dt=dt, dx=dx, num_steps=steps), O.svhn_spec(N, C, dt, dx, steps)
    if kind == "cifar2":
        return (quiet(P.LearnableDiffusionLayer, N, C, dt=dt, dx=dx, dy=dx, num_steps=steps),
                O.cifar2_spec(N, C, dt, dx, dx, steps))
    layer = quiet(P.EnhancedDiffusionLayer, N, C, dt=dt, dx=dx, dy=dx, num_steps=steps,
                  channel_mixing_enabled=(kind != "plain"))
    spec = O.cifar10_spec(N, C, dt, dx, dx, steps)
    if kind == "plain":
        spec = O.AdiSpec(N, C, dt, dx, dx, steps, "strang", False, 10.0, "none", False)
    return layer, spec


# Three cases from wider seeded walks of round 3 (seeds 2026 and 99, 889 cases each) whose few-entry gradients fall
# outside the window above.  Round 4 carried every stage of those sums in double precision (pde_blend.hip, the epilogue of
# the C <= 4 kernels, adi_pgrad_kernel) and the figures did not move (2.9e-5 -> 2.8e-5, 1.6e-4 -> 1.6e-4, 2.06e-5 ->
# 2.12e-5): the summation is not what limits them.  Each is sigma'(w) * sum g.(u0 - u_K), or sum g.u of a 1 x 1 operator, of
# a layer whose output differs from its input by ~1e-3: the scalar inherits the fp32 rounding of the STATES (the two-sided
# elimination multiplies by stored reciprocals where the reference divides: about one rounding more per sweep), amplified
# by the cancellation — 3-8 times the reference's own fp32 distance from the fp64 value on these three.  They are pinned
# here with the bounds they meet, so that a regression (or an improvement of the sweeps' rounding) shows.
PINNED = [
    (("svhn", 12, 8, 1, 0.02, 2.0, 1.0, 5.0, 17), 4e-5),       # skip weight: 2.8e-5 from fp64 (reference's fp32: 7.0e-6)
    (("svhn", 28, 16, 3, 0.1, 2.0, 1.0, 0.0, 1), 2.5e-4),       # skip weight: 1.6e-4 (1.9e-5); B = 1, u_K within 2e-3 of u0
    (("cifar10", 16, 1, 3, 0.4, 1.0, 1.0, 0.0, 1), 3e-5),       # 1 x 1 operator: 2.1e-5 (1.8e-6)
    # round 4, seed 31415 (889 cases): the same scalar once more, through the one-launch C <= 4 kernel
    (("svhn", 8, 3, 4, 0.02, 1.0, 1.0, 0.0, 40), 4e-4),         # skip weight: 2.5e-4 (7.6e-5): 8 x 8 planes, u_K within 1e-3 of u0
    (("svhn", 24, 1, 4, 0.02, 2.0, 0.3, 0.3, 17), 6e-5),        # skip weight 4.4e-5 (2.4e-6), 1 x 1 coupling 2.4e-5 (3.9e-7; 3.6e-5
                                                                #   before the per-thread sums of mix_gm_kernel went to double)
    (("svhn", 32, 3, 1, 0.1, 2.0, 0.3, 5.0, 17), 1e-4),         # skip weight 7.9e-5 (1.8e-5): 4.5 x the reference's distance
]


@pytest.mark.parametrize("case,limit", PINNED, ids=lambda c: "-".join(str(x) for x in c) if isinstance(c, tuple) else str(c))
def test_pinned_cancelling_scalar_gradients(case, limit):
    check_case(case, small_limit=limit)


def _cases():
    rng = random.Random(SEED)
    return [_draw(rng) for _ in range(CASES)]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "-".join(str(x) for x in c))
def test_random_case_vs_oracle(case):
    check_case(case)


def check_case(case, small_limit=None):
    """One drawn layer call against the oracle (also used by tests/test_gpu_anysize.py for the other line lengths).
    ``small_limit``: the pinned cases' own bound on the few-entry gradients (see PINNED)."""
    kind, N, C, steps, dt, dx, scale, slope, B = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    layer, spec = _build(kind, N, C, steps, dt, dx)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(scale * (1 + 0.2 * torch.randn(p.shape, generator=g)))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(slope * torch.randn(p.shape, generator=g))
            elif n in ("channel_mixing", "channel_coupling"):
                p.copy_(torch.eye(C) + (0.3 / C ** 0.5) * torch.randn(C, C, generator=g))
            elif n == "skip_weight":
                p.fill_(float(torch.randn(1, generator=g)))
    if kind == "plain":
        layer.channel_mixing.requires_grad_(False)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    params = {k: v.detach().clone() for k, v in layer.named_parameters() if v.requires_grad}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    dl = layer.cuda()
    ud = u.cuda().requires_grad_(True)
    y = dl(ud)
    y.backward(gy.cuda())
    errs = {"y": G.rel_err(y.detach().cpu(), y_ref), "gu": G.rel_err(ud.grad.cpu(), gu_ref)}
    for n, p in dl.named_parameters():
        if p.requires_grad:
            errs["g_" + n] = G.rel_err(p.grad.cpu().reshape(gp_ref[n].shape), gp_ref[n])
    # Parameters of a handful of entries (skip_weight; the operator at C <= 2) are sums of B*C*N*N signed terms that
    # largely cancel (seen: 2880 terms of total size 220 adding up to 0.0144), so fp32 rounding noise relative to the SUM is
    # far above 1e-5 for the reference's own arithmetic too: on that case the fp32 oracle is 1.2e-4 away from the same
    # restatement evaluated in fp64, the HIP result 1.9e-4.  Those gradients are therefore decided against the fp64 oracle
    # (.double() inputs): within 2e-5 of it, or no noisier than 4x the fp32 oracle's own distance from it.  Everything
    # else against the fp32 oracle at 1e-5.
    small = {n for n, v in gp_ref.items() if v.numel() <= 4}
    limits = {}
    if small:
        p64 = {k: v.double() for k, v in params.items()}
        _, _, gp64 = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u.double(), p64, gy.double())
        for n in small:
            errs["g_" + n] = G.rel_err(dl.get_parameter(n).grad.cpu().reshape(gp64[n].shape), gp64[n])
            o32 = G.rel_err(gp_ref[n], gp64[n])
            errs["oracle32_vs_64_" + n] = o32                                     # reported with a failure
            # (the reference's own fp32 distance enters with at most 1e-4: a noisy fp32 oracle cannot open the window beyond
            #  4e-4, and the window never falls below the distance the reference's own arithmetic is at)
            limits["g_" + n] = small_limit if small_limit is not None else max(2e-5, 4.0 * min(o32, 1e-4))
            limits["oracle32_vs_64_" + n] = float("inf")
    bad = {k: (v, limits.get(k, 1e-5)) for k, v in errs.items() if not v <= limits.get(k, 1e-5)}
    assert not bad, (bad, {k: v for k, v in errs.items() if k.startswith("oracle32")})


def _explicit_cases():
    rng = random.Random(SEED + 1)
    out = []
    for _ in range(max(8, CASES // 3)):
        out.append((rng.choice([8, 12, 16, 20, 24, 32, 36, 48, 64]), rng.choice([1, 2, 3, 5, 8]), rng.randint(1, 4),
                    rng.choice([1, 2, 5, 11, 30]), rng.choice([0.01, 0.5, 2.0]), rng.choice(["f32", "f32", "bf16"])))
    return out


@pytest.mark.parametrize("case", _explicit_cases(), ids=lambda c: "-".join(str(x) for x in c))
def test_random_explicit_case_vs_oracle(case):
    """tiny_imagenet.ImprovedDiffusionLayer at random plane sizes (wave-per-plane kernels at 16/32/64, the generic
    kernel elsewhere), channel counts, step counts and batch sizes (ragged plane counts), fp32 and bf16 tensors."""
    import cnn_with_pde_amd as P
    size, C, steps, B, dt, dtn = case
    dtype = torch.float32 if dtn == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    layer = P.ImprovedDiffusionLayer(size, C, dt=dt, num_steps=steps)
    with torch.no_grad():
        layer.alpha_base.copy_(0.25 * torch.rand(C, generator=g))           # some beyond the 0.15 clamp, some tiny
        layer.channel_scaling.copy_(1 + 0.3 * torch.randn(C, generator=g))
    u = torch.randn(B, C, size, size, generator=g).to(dtype).float()
    gy = torch.randn(B, C, size, size, generator=g).to(dtype).float()
    params = {k: v.detach().clone() for k, v in layer.named_parameters() if k != "beta_base"}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.tiny_forward(a, p, dt=dt, num_steps=steps), u, params, gy)
    dl = layer.cuda()
    ud = u.to(dtype).cuda().requires_grad_(True)
    y = dl(ud)
    y.backward(gy.to(dtype).cuda())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    errs = {"y": G.rel_err(y.detach().float().cpu(), y_ref), "gu": G.rel_err(ud.grad.float().cpu(), gu_ref)}
    for n in params:
        errs["g_" + n] = G.rel_err(getattr(dl, n).grad.float().cpu(), gp_ref[n])
    # bf16 tensors: the per-channel parameter gradients are sums over the whole batch that partly cancel, and outside
    # the plane sizes with a fused time loop the state makes a bf16 round trip per step which the oracle does not model
    ptol = tol if dtype == torch.float32 else 1e-1
    limits = {k: (ptol if k.startswith("g_") else tol) for k in errs}
    if dtype == torch.float32:
        # gradients of at most four entries (C <= 4 channels: sums over the whole batch that cancel): decided against the
        # oracle in fp64, within 2e-5 or no noisier than 4x the fp32 oracle itself — see test_random_case_vs_oracle
        small = [n for n in params if gp_ref[n].numel() <= 4]
        if small:
            _, _, gp64 = O.value_and_grads(lambda a, p: O.tiny_forward(a, p, dt=dt, num_steps=steps), u.double(),
                                           {k: v.double() for k, v in params.items()}, gy.double())
            for n in small:
                errs["g_" + n] = G.rel_err(getattr(dl, n).grad.float().cpu(), gp64[n])
                limits["g_" + n] = max(2e-5, 4.0 * min(G.rel_err(gp_ref[n], gp64[n]), 1e-4))
    bad = {k: (v, limits[k]) for k, v in errs.items() if not v <= limits[k]}
    assert not bad, (bad, errs)


def _bf16_cases():
    rng = random.Random(SEED + 2)
    out = []
    for _ in range(max(8, CASES // 4)):
        kind = rng.choice(["cifar10", "svhn", "plain"])
        out.append((kind, rng.choice([16, 28, 32]), rng.choice([2, 3, 6, 32, 64, 128]), rng.randint(1, 3),
                    rng.choice([0.01, 0.1]), rng.choice([1, 2, 4])))
    return out


@pytest.mark.parametrize("case", _bf16_cases(), ids=lambda c: "-".join(str(x) for x in c))
def test_random_bf16_case_vs_oracle(case):
    """bf16 tensors (fp32 arithmetic inside; bf16 MFMA operators at C = 64 / 128) against the oracle rounding its
    state where the layer stores it (state_cast): 1e-2, and 3e-2 against the plain fp32 oracle."""
    kind, N, C, steps, dt, B = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    layer, spec = _build(kind, N, C, steps, dt, 1.0)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif n in ("channel_mixing", "channel_coupling"):
                p.copy_(torch.eye(C) + (0.3 / C ** 0.5) * torch.randn(C, C, generator=g))
    if kind == "plain":
        layer.channel_mixing.requires_grad_(False)
    u = torch.randn(B, C, N, N, generator=g).bfloat16().float()
    gy = torch.randn(B, C, N, N, generator=g).bfloat16().float()
    params = {k: v.detach().clone() for k, v in layer.named_parameters() if v.requires_grad}
    cast = lambda t: t.bfloat16().float()
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec, cast), u, params, gy)
    dl = layer.cuda()
    ud = u.bfloat16().cuda().requires_grad_(True)
    y = dl(ud)
    assert y.dtype == torch.bfloat16
    y.backward(gy.bfloat16().cuda())
    errs = {"y": G.rel_err(y.detach().float().cpu(), y_ref), "gu": G.rel_err(ud.grad.float().cpu(), gu_ref)}
    for n, p in dl.named_parameters():
        if p.requires_grad and n != "skip_weight":
            errs["g_" + n] = G.rel_err(p.grad.float().cpu().reshape(gp_ref[n].shape), gp_ref[n])
    if "skip_weight" in gp_ref:
        # one scalar = a sum of B*C*N*N signed terms that nearly cancel for random gy: in bf16 the rounding noise of the
        # terms can exceed the sum itself (fp32 tensors are held to 1e-5 in test_random_case_vs_oracle); measured
        # against the size of the neighbouring coupling gradient instead of against itself
        scale = max(abs(float(gp_ref["skip_weight"])), 0.25 * float(gp_ref["channel_coupling"].abs().max()))
        errs["g_skip_weight"] = abs(float(dl.skip_weight.grad) - float(gp_ref["skip_weight"])) / scale
    bad = {k: v for k, v in errs.items() if not v <= 2e-2}
    assert not bad, (bad, errs)


def _extractor_cases():
    rng = random.Random(SEED + 3)
    return [(rng.choice([16, 28, 32]), rng.choice([1, 2, 3, 4]), rng.choice([1, 2, 5, 33, 130, 1100]), rng.random() < 0.5)
            for _ in range(max(6, CASES // 5))]


@pytest.mark.parametrize("case", _extractor_cases(), ids=lambda c: "-".join(str(x) for x in c))
def test_random_extractor_fused_vs_layer_by_layer(case):
    """cifar10.MultiScaleExtractor's counterpart (three layers on one input, attention pool / gate / combination) with
    the one-launch group + fused epilogue against one call per layer + torch epilogue, at random sizes and batches
    (side-by-side and one-after-the-other arrangements), values and every gradient."""
    import copy
    import cnn_with_pde_amd as P
    N, C, B, feats = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    torch.manual_seed(zlib.crc32(repr(case).encode()) & 0x7FFFFFFF)
    m1 = quiet(P.MultiScaleExtractor, N, C).cuda()
    with torch.no_grad():
        for n, p in m1.named_parameters():
            if n.endswith("alpha_base") or n.endswith("beta_base"):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=g).cuda())
            elif n.endswith("channel_mixing"):
                p.copy_((torch.eye(C) + 0.1 * torch.randn(C, C, generator=g)).cuda())
        m1.combine_weights.copy_(torch.randn(3, generator=g))
    m1.return_features = feats
    m2 = copy.deepcopy(m1)
    m2.fused_epilogue = False
    for ly in (m2.pde1, m2.pde2, m2.pde3):
        ly.small_channel_kernels = False             # per-step launches, one layer at a time: nothing shared with m1's path
    x = torch.randn(B, C, N, N, generator=g).cuda()
    gy = torch.randn(B, C, N, N, generator=g).cuda()
    res = []
    for m in (m1, m2):
        xd = x.clone().requires_grad_(True)
        out = m(xd)
        loss = (out[0] * gy).sum()
        if feats and m is m1 or (feats and out[1] is not None):
            loss = loss + sum((f * gy).sum() * 0.3 for f in out[1:] if f is not None)
        loss.backward()
        res.append([out[0].detach(), xd.grad] + [p.grad for _, p in sorted(m.named_parameters())])
    scale = max(float(t.abs().max()) for t in res[1][2:])
    for a, b, (n, _) in zip(res[0], res[1], [("out", 0), ("gx", 0)] + sorted(m1.named_parameters())):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3 * scale if n not in ("out", "gx") else 1e-30)
        assert err <= 1e-4, (n, err)
