"""GPU parity of the hand-scheduled assembly backward kernel (csrc/gen_adi_bwd_asm.py) — the adjoint of the reference's
Strang time loop (mnist_test.py:44-65, cifar10.py:74-114) at N = 32 on fp32 tensors.

The kernel is what `pde_adi_backward` runs by default for such schedules (`pde_adi_backward_kernel` says so); these tests
hold it to the CPU oracle at 1e-5 (max-norm relative, the tolerance north_star states) on seeded inputs that cover what
its control flow branches on — ragged batches (planes beyond the batch inside the last pass), a batch smaller than one pass,
more groups than passes, channel counts with and without the XCD-ordered grid, time-dependent coefficients (the
time-weighted sums and their chunk-to-chunk summation by parts), channels whose clamp mask moves in time (those stay with
the masked HIP body inside the same call) — and to the HIP kernel it replaces (child interpreter with PDE_ASM_BWD=0):
input gradients bitwise equal, parameter gradients within 2e-6."""
import os
import subprocess
import sys

import pytest
import torch

import golden_util as G
from oracle import pde_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
NAMES = ["alpha_base", "beta_base", "alpha_time_coeff", "beta_time_coeff"]
CASES = [  # B, C, steps, dt, relative spread, slope scale
    (37, 5, 3, 0.02, 0.15, 1.0),       # ragged batch, odd channel count
    (3, 8, 2, 0.05, 0.15, 0.5),        # fewer samples than one workgroup pass; XCD-ordered grid
    (100, 16, 4, 0.01, 0.10, 0.0),     # several passes per group, no time dependence
    (1, 1, 2, 0.01, 0.10, 0.3),        # one plane
    (70, 3, 5, 0.004, 0.20, 2.0),      # strong time dependence: every time-weighted sum matters
]


def _inputs(ci):
    B, C, steps, dt, rel, slope = CASES[ci]
    g = torch.Generator().manual_seed(500 + ci)
    N = 32
    spec = O.cifar10_spec(N, C, dt=dt, num_steps=steps)
    params = O.adi_init_params(spec, "cifar10", gen=g)
    params["channel_mixing"] = torch.eye(C)
    for k in ("alpha_base", "beta_base"):
        params[k] = params[k] * (1 + rel * torch.randn(params[k].shape, generator=g))
    for k in ("alpha_time_coeff", "beta_time_coeff"):
        params[k] = slope * torch.randn(params[k].shape, generator=g)
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    return spec, params, u, gy, steps, dt


def _run_gpu(params, u, gy, steps, dt):
    import cnn_with_pde_amd as P
    sweeps = [s for st in P.adi_schedule(dt, 1.0, 1.0, steps) for s in st]
    ud = u.cuda().requires_grad_(True)
    ps = [params[k].cuda().requires_grad_(True) for k in NAMES]
    y = P.adi_diffuse(ud, *ps, sweeps, smooth3=False, clamp_max=10.0, checkpoints=0)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    return y.detach().cpu(), ud.grad.cpu(), {k: p.grad.cpu() for k, p in zip(NAMES, ps)}


def _desc(B, C, N, steps, dt, split="strang", io=None):
    import cnn_with_pde_amd as P
    import cnn_with_pde_amd.functional as F
    import cnn_with_pde_amd._lib as L
    sweeps = [s for st in P.adi_schedule(dt, 1.0, 1.0, steps, split) for s in st]
    return F._build_desc(B, C, N, L.PDE_IO_F32 if io is None else io, sweeps, False, 10.0, 1e-6)


def test_the_assembly_kernel_is_what_runs():
    import cnn_with_pde_amd._lib as L
    import ctypes as C
    lib = L.load()
    if os.environ.get("PDE_ASM_BWD", "1")[0] == "0":
        pytest.skip("assembly kernel switched off in this environment")
    q = lambda d, nck=0: lib.pde_adi_backward_kernel(C.byref(d), nck)
    assert q(_desc(512, 64, 32, 10, 0.001)) == 1                 # the headline schedule
    assert q(_desc(3, 5, 32, 2, 0.01)) == 1
    assert q(_desc(512, 64, 32, 10, 0.001), 2) == 0              # checkpoints: HIP kernel
    assert q(_desc(64, 1, 28, 10, 0.001)) == 0                   # N = 28
    assert q(_desc(64, 3, 32, 1, 0.001)) == 0                    # one step (the per-step launches)
    assert q(_desc(64, 3, 32, 4, 0.001, "lie")) == 0             # Lie steps (cifar_2version.py:93-99)
    assert q(_desc(64, 3, 32, 4, 0.001, io=L.PDE_IO_BF16)) == 0  # bf16 tensors
    assert q(_desc(8, 2, 40, 4, 0.001)) == 2                     # any-size path


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_against_the_oracle(ci):
    spec, params, u, gy, steps, dt = _inputs(ci)
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    y, gu, gp = _run_gpu(params, u, gy, steps, dt)
    errs = {"y": G.rel_err(y, y_ref), "gu": G.rel_err(gu, gu_ref)}
    errs.update({k: G.rel_err(gp[k], gp_ref[k]) for k in NAMES})
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)


def test_channels_with_moving_clamp_masks_share_the_call():
    """Two of four channels get coefficients that cross the clamp floor during the time window: the factor kernel flags
    them on the device, the assembly kernel leaves them alone and the masked HIP body (launched right behind it over the
    same groups) owns them; all four channels must match the oracle."""
    g = torch.Generator().manual_seed(77)
    N, C, B, steps, dt = 32, 4, 21, 3, 0.05
    spec = O.cifar10_spec(N, C, dt=dt, num_steps=steps)
    params = O.adi_init_params(spec, "cifar10", gen=g)
    params["channel_mixing"] = torch.eye(C)
    for k in ("alpha_base", "beta_base"):
        params[k] = params[k] * (1 + 0.1 * torch.randn(params[k].shape, generator=g))
    for k in ("alpha_time_coeff", "beta_time_coeff"):
        params[k] = 0.2 * torch.randn(params[k].shape, generator=g)
    # channels 1 and 3: base + slope * t crosses eps inside [0, steps * dt] at some points
    for c in (1, 3):
        params["alpha_base"][c, ::3, ::2] = 0.02
        params["alpha_time_coeff"][c, ::3, ::2] = -0.3
        params["beta_base"][c, 1::4, :] = 0.01
        params["beta_time_coeff"][c, 1::4, :] = -0.2
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
    y, gu, gp = _run_gpu(params, u, gy, steps, dt)
    errs = {"y": G.rel_err(y, y_ref), "gu": G.rel_err(gu, gu_ref)}
    errs.update({k: G.rel_err(gp[k], gp_ref[k]) for k in NAMES})
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)


CHILD = r"""
import sys, torch
sys.path[:0] = [%(root)r, %(tests)r]
import test_gpu_asm_bwd as T
out = {}
for ci in range(len(T.CASES)):
    spec, params, u, gy, steps, dt = T._inputs(ci)
    y, gu, gp = T._run_gpu(params, u, gy, steps, dt)
    out[ci] = (y, gu, gp)
torch.save(out, %(path)r)
"""


def test_against_the_hip_kernel_it_replaces(tmp_path):
    """Same inputs through both kernels, each in its own interpreter (the switch is read once per process)."""
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    # (the "asm" child also takes the opt-in assembly FORWARD, gen_adi_fwd_asm.py: its output must equal the HIP forward's
    #  bit for bit — the `torch.equal(y0, y1)` below)
    for tag, env in (("hip", {"PDE_ASM_BWD": "0", "PDE_ASM_FWD": "0"}), ("asm", {"PDE_ASM_BWD": "1", "PDE_ASM_FWD": "1"})):
        path = str(tmp_path / f"{tag}.pt")
        code = CHILD % {"root": os.path.dirname(here), "tests": here, "path": path}
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        res[tag] = torch.load(path, weights_only=True)
    for ci in res["hip"]:
        (y0, gu0, gp0), (y1, gu1, gp1) = res["hip"][ci], res["asm"][ci]
        assert torch.equal(y0, y1), ci                         # (the forward is the same kernel)
        assert torch.equal(gu0, gu1), (ci, G.rel_err(gu1, gu0))    # same adjoint arithmetic, instruction for instruction
        for k in NAMES:
            assert G.rel_err(gp1[k], gp0[k]) <= 2e-6, (ci, k, G.rel_err(gp1[k], gp0[k]))


def test_bitwise_repeatable():
    spec, params, u, gy, steps, dt = _inputs(0)
    a = _run_gpu(params, u, gy, steps, dt)
    b = _run_gpu(params, u, gy, steps, dt)
    assert torch.equal(a[1], b[1])
    for k in NAMES:
        assert torch.equal(a[2][k], b[2][k]), k
