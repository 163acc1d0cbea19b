#!/usr/bin/env python3
"""Diagnostic: one drawn case of tests/test_gpu_fuzz.py under several checkpoint policies, every gradient against the fp32
and the fp64 oracle.  usage: diag_fuzz_case.py kind N C steps dt dx scale slope B"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import golden_util as G  # noqa: E402
import test_gpu_fuzz as Z  # noqa: E402
from oracle import pde_oracle as O  # noqa: E402

a = sys.argv[1:]
case = (a[0], int(a[1]), int(a[2]), int(a[3]), float(a[4]), float(a[5]), float(a[6]), float(a[7]), int(a[8]))
kind, N, C, steps, dt, dx, scale, slope, B = case
for ck in ("auto", 0, "all"):
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    layer, spec = Z._build(kind, N, C, steps, dt, dx)
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
    u = torch.randn(B, C, N, N, generator=g)
    gy = torch.randn(B, C, N, N, generator=g)
    params = {k: v.detach().clone() for k, v in layer.named_parameters() if v.requires_grad}
    _, gu32, gp32 = O.value_and_grads(lambda x, p: O.adi_forward(x, p, spec), u, params, gy)
    _, gu64, gp64 = O.value_and_grads(lambda x, p: O.adi_forward(x, p, spec), u.double(),
                                      {k: v.double() for k, v in params.items()}, gy.double())
    sps = 3 if spec.split == "strang" else 2
    layer.checkpoint_policy = ck if ck != "all" else (1 << (sps - 1)) - 1 if spec.mix != "none" else (1 << (sps * steps - 1)) - 1
    dl = layer.cuda()
    ud = u.cuda().requires_grad_(True)
    dl(ud).backward(gy.cuda())
    print("policy", ck, "->", dl.checkpoint_policy)
    print("   gu: vs64 %.2e  (oracle32 vs64 %.2e)" % (G.rel_err(ud.grad.cpu().double(), gu64), G.rel_err(gu32.double(), gu64)))
    for n, p in dl.named_parameters():
        if p.requires_grad:
            print("   %-18s vs64 %.2e  (oracle32 vs64 %.2e)" % (n, G.rel_err(p.grad.cpu().double().reshape(gp64[n].shape), gp64[n]),
                                                                G.rel_err(gp32[n].double(), gp64[n])))
