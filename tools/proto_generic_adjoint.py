#!/usr/bin/env python3
"""Prototype (CPU, numpy loops) of the any-line-length ADI kernels of csrc/pde_adi_gen.hip: the same per-line recurrences
in the same order, checked against the oracle's autograd.  Development aid only (the product is the HIP code).
usage: proto_generic_adjoint.py [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pde_oracle as O  # noqa: E402


def factor_line(base, slope, t, delta, h2, eps, cmax, smooth3):
    n = len(base)
    th = base + slope * np.float32(t)
    ok = (th >= eps) & ((th <= cmax) if cmax is not None else True)
    th = np.clip(th, eps, cmax if cmax is not None else None).astype(np.float32)
    if smooth3:
        p = np.concatenate([th[:1], th, th[-1:]])
        th = ((p[:-2] + p[1:-1] + p[2:]) / np.float32(3)).astype(np.float32)
    co = (th * np.float32(delta) / np.float32(h2)).astype(np.float32)
    cs = np.zeros(n, np.float32)
    inv = np.zeros(n, np.float32)
    for i in range(n):
        b = 1 + (1 if i in (0, n - 1) else 2) * co[i]
        den = b + (co[i] * cs[i - 1] if i else 0) + eps
        inv[i] = 1 / den
        cs[i] = -co[i] / den if i < n - 1 else 0
    return co, cs, inv, ok.astype(np.float32)


def solve_line(d, co, cs, inv):
    n = len(d)
    x = d.copy()
    x[0] = x[0] * inv[0]
    for i in range(1, n):
        x[i] = (x[i] + co[i] * x[i - 1]) * inv[i]
    for i in range(n - 2, -1, -1):
        x[i] = x[i] - cs[i] * x[i + 1]
    return x


def adjoint_line(r, x, co, cs, inv, ok, scale, eps, smooth3):
    """r: incoming adjoint (dL/dx_new), x: the sweep's output.  Returns (dL/dx_old, x_old, dL/dtheta (masked))."""
    n = len(r)
    w = r.copy()
    for k in range(1, n):
        w[k] = w[k] - cs[k - 1] * w[k - 1]
    lam = w
    lam[n - 1] = lam[n - 1] * inv[n - 1]
    for k in range(n - 2, -1, -1):
        lam[k] = (lam[k] + co[k + 1] * lam[k + 1]) * inv[k]
    gsm = np.zeros(n, np.float32)
    xo = x.copy()
    for k in range(n):
        q = (1 if k in (0, n - 1) else 2) * x[k] - (x[k - 1] if k else 0) - (x[k + 1] if k < n - 1 else 0)
        gsm[k] = -lam[k] * q * scale
        xo[k] = (1 + eps) * x[k] + co[k] * q
    if smooth3:
        p = np.concatenate([[0], gsm, [0]]).astype(np.float32)
        gth = (p[:-2] + p[1:-1] + p[2:]) / np.float32(3)
        gth[0] += gsm[0] / np.float32(3)
        gth[-1] += gsm[-1] / np.float32(3)
    else:
        gth = gsm
    return lam, xo, gth * ok


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.manual_seed(3)
    spec = O.AdiSpec(N, 2, 0.05, 1.0, 1.5, 2, "strang", True, 2.2, "none", False)
    g = torch.Generator().manual_seed(5)
    params = {k: v.clone() for k, v in O.adi_init_params(spec, "cifar10", gen=g).items() if k.startswith(("alpha", "beta"))}
    params["alpha_base"] = 2.0 + 0.5 * torch.randn(2, N, N, generator=g)      # some entries beyond clamp_max
    params["beta_base"] = 1.0 + 0.8 * torch.randn(2, N, N, generator=g)       # some below eps
    params["alpha_time_coeff"] = torch.randn(2, N, N, generator=g)
    params["beta_time_coeff"] = torch.randn(2, N, N, generator=g)
    u = torch.randn(3, 2, N, N, generator=g)
    gy = torch.randn(3, 2, N, N, generator=g)
    y, gu, gp = O.value_and_grads(lambda uu, pp: O.adi_forward(uu, pp, spec), u, params, gy)
    sched = O.sweep_schedule(spec)
    B, C = u.shape[:2]
    P = {k: v.numpy() for k, v in params.items()}
    fac = []
    for axis, delta, t in sched:
        base, slope = (P["alpha_base"], P["alpha_time_coeff"]) if axis == 0 else (P["beta_base"], P["beta_time_coeff"])
        h2 = spec.dx ** 2 if axis == 0 else spec.dy ** 2
        f = np.zeros((C, N, 4, N), np.float32)
        for c in range(C):
            for ln in range(N):
                bl, sl = (base[c, ln, :], slope[c, ln, :]) if axis == 0 else (base[c, :, ln], slope[c, :, ln])
                f[c, ln] = factor_line(bl, sl, t, delta, h2, spec.eps, spec.clamp_max, spec.smooth3)
        fac.append(f)
    X = u.numpy().copy()
    for (axis, delta, t), f in zip(sched, fac):
        for b in range(B):
            for c in range(C):
                for ln in range(N):
                    d = X[b, c, ln, :] if axis == 0 else X[b, c, :, ln]
                    d[:] = solve_line(d.copy(), f[c, ln, 0], f[c, ln, 1], f[c, ln, 2])
    print("forward ", np.abs(X - y.numpy()).max() / np.abs(y.numpy()).max())
    R = gy.numpy().copy()
    G = {k: np.zeros_like(v) for k, v in P.items()}
    for (axis, delta, t), f in list(zip(sched, fac))[::-1]:
        h2 = spec.dx ** 2 if axis == 0 else spec.dy ** 2
        nb, ns = ("alpha_base", "alpha_time_coeff") if axis == 0 else ("beta_base", "beta_time_coeff")
        for b in range(B):
            for c in range(C):
                for ln in range(N):
                    r = R[b, c, ln, :] if axis == 0 else R[b, c, :, ln]
                    x = X[b, c, ln, :] if axis == 0 else X[b, c, :, ln]
                    lam, xo, gth = adjoint_line(r.copy(), x.copy(), f[c, ln, 0], f[c, ln, 1], f[c, ln, 2], f[c, ln, 3],
                                                np.float32(delta) / np.float32(h2), spec.eps, spec.smooth3)
                    r[:] = lam
                    x[:] = xo
                    gb = G[nb][c, ln, :] if axis == 0 else G[nb][c, :, ln]
                    gs = G[ns][c, ln, :] if axis == 0 else G[ns][c, :, ln]
                    gb += gth
                    gs += np.float32(t) * gth
    print("gu      ", np.abs(R - gu.numpy()).max() / np.abs(gu.numpy()).max())
    print("rebuilt ", np.abs(X - u.numpy()).max() / np.abs(u.numpy()).max())
    for k in G:
        print(k, np.abs(G[k] - gp[k].numpy()).max() / np.abs(gp[k].numpy()).max())


if __name__ == "__main__":
    main()
