#!/usr/bin/env python3
"""Diagnostic: forward + backward time of one implicit layer at line lengths with and without fused kernels
(pde_adi_line_length_path 1 / 2).  usage: perf_anysize.py [B] [C]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cnn_with_pde_amd as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for N in (32, 36, 48, 64, 96, 128):
    with contextlib.redirect_stdout(io.StringIO()):
        ly = P.EnhancedDiffusionLayer(N, C, dt=0.01, num_steps=10, channel_mixing_enabled=False).cuda()
    u = torch.randn(B, C, N, N, device="cuda", requires_grad=True)
    gy = torch.randn_like(u)

    def step():
        for p in ly.parameters():
            p.grad = None
        u.grad = None
        ly(u).backward(gy)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"N={N:4d} ({B}x{C} planes, 30 sweeps): {ms:8.3f} ms fwd+bwd, {B * C * N * N * 30 / ms / 1e6:8.1f} M element-sweeps/ms", flush=True)
