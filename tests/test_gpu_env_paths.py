"""GPU parity of the paths an environment switch selects (each switch is read once per process, so the cases run in
a child interpreter):

    PDE_KMAX_MAPPED=0   the per-sweep coefficient maxima reach the host through a device buffer and an asynchronous
                        copy instead of the factor epilogue writing straight into the caller's pinned buffer — the
                        path every process takes that sees more than one GPU (the 8-GPU scaling run)
    PDE_WIDE=0          layers with a channel operator at C = 32 / 64 on the per-step launches (the one-launch forward
                        of pde_adi_wide.h off)
    PDE_MIX_NO_SPLIT=1  the channel operator's backward on fp32 tensors at C = 64 on the fp32 MFMA (mix_bwd_fused_kernel)
                        instead of the three-piece bf16 products that are the default since round 3
    PDE_RH_NO_SPLIT=1   the symmetric layer's gradient of K on the fp32 MFMA (rh_outer_kernel) instead of the three-piece
                        bf16 kernel

Each child runs a slice of the ordinary parity tests (golden vectors whose backward plans checkpoints from those
maxima; the BASELINE configurations as layers against the oracle)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _child(env_extra, args):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + args,
                       cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    return r.stdout


def test_coefficient_maxima_through_the_copy_path():
    out = _child({"PDE_KMAX_MAPPED": "0"},
                 ["tests/test_gpu_golden.py", "-k", "fashion or mnist_trained or svhn_live or cifar10_default",
                  "tests/test_gpu_small.py::test_small_kernels_large_coefficients_checkpoints"])
    assert "deselected" in out or "passed" in out


def test_per_step_launches_with_the_one_launch_forward_off():
    _child({"PDE_WIDE": "0"},
           ["tests/test_gpu_configs.py", "-k", "cfg2 or cfg3 or per_step"])


def test_fp32_mfma_kernels_behind_the_three_piece_products():
    _child({"PDE_MIX_NO_SPLIT": "1", "PDE_RH_NO_SPLIT": "1"},
           ["tests/test_gpu_configs.py", "-k", "cfg2", "tests/test_gpu_parity.py::test_channel_mix_vs_fp64",
            "tests/test_gpu_rh.py", "-k", "cfg2 or channel_mix_vs_fp64 or rh"])
