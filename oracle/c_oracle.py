"""ctypes front end of oracle/libpde_oracle_c.so (C restatement with closed-form gradients).
TEST INFRASTRUCTURE — see the header of pde_oracle_c.c."""
import ctypes as C
import os

import numpy as np
import torch

from . import pde_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpde_oracle_c.so")


class OSweep(C.Structure):
    _fields_ = [("axis", C.c_int), ("delta", C.c_double), ("h2", C.c_double), ("t", C.c_double)]


def available() -> bool:
    return os.path.isfile(LIB)


def _sweeps(spec: O.AdiSpec):
    sch = O.sweep_schedule(spec)
    arr = (OSweep * len(sch))()
    for i, (ax, delta, t) in enumerate(sch):
        h = spec.dx if ax == 0 else spec.dy
        arr[i].axis, arr[i].delta, arr[i].h2, arr[i].t = ax, delta, h ** 2, t
    return arr, len(sch)


def adi_value_and_grads(u, params, gy, spec: O.AdiSpec):
    """(y, gu, {four parameter grads}) for a spec WITHOUT channel mixing / skip, fp32 or fp64."""
    assert spec.mix == "none" and not spec.skip
    lib = C.CDLL(LIB)
    dt = u.dtype
    suf = "_f64" if dt == torch.float64 else "_f32"
    npdt = np.float64 if dt == torch.float64 else np.float32
    B, Cc, N, _ = u.shape
    chw = lambda p: np.ascontiguousarray((p if p.dim() == 3 else p.unsqueeze(0)).detach().numpy().astype(npdt))
    ab, bb, as_, bs = (chw(params[k]) for k in ("alpha_base", "beta_base", "alpha_time_coeff", "beta_time_coeff"))
    un = np.ascontiguousarray(u.detach().numpy().astype(npdt))
    gn = np.ascontiguousarray(gy.detach().numpy().astype(npdt))
    y = np.empty_like(un)
    gu = np.empty_like(un)
    gr = [np.empty_like(ab) for _ in range(4)]
    sw, S = _sweeps(spec)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    common = (B, Cc, N, S, sw, int(spec.smooth3), int(spec.clamp_max is not None),
              C.c_double(spec.clamp_max or 0.0), C.c_double(spec.eps), p(ab), p(bb), p(as_), p(bs))
    rc = getattr(lib, "oracle_adi_forward" + suf)(*common, p(un), p(y), None)
    assert rc == 0
    rc = getattr(lib, "oracle_adi_backward" + suf)(*common, p(un), p(gn), p(gu), *[p(g) for g in gr])
    assert rc == 0
    names = ("alpha_base", "beta_base", "alpha_time_coeff", "beta_time_coeff")
    return torch.from_numpy(y), torch.from_numpy(gu), {k: torch.from_numpy(g).reshape(params[k].shape)
                                                       for k, g in zip(names, gr)}
