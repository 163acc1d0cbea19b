#!/usr/bin/env python3
"""Diagnostic: the channel-operator backward on fp32 tensors at C = 64 — bf16 three-piece products (default) against the
fp32 MFMA kernel (PDE_MIX_NO_SPLIT=1 in a child process), both against fp64."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import cnn_with_pde_amd as P
    import golden_util as G
    for seed, B, HW, scale in ((1, 5, 1024, 0.1), (2, 64, 1024, 0.3), (3, 7, 64, 1.0)):
        g = torch.Generator().manual_seed(seed)
        C = 64
        u = torch.randn(B, C, HW, generator=g)
        M = torch.eye(C) + scale * torch.randn(C, C, generator=g)
        go = torch.randn(B, C, HW, generator=g) * torch.logspace(-3, 3, C).view(1, C, 1)       # wide dynamic range over channels
        ud, Md = u.cuda().requires_grad_(True), M.cuda().requires_grad_(True)
        out = P.channel_mix(ud.view(B, C, HW, 1), Md)
        out.backward(go.view(B, C, HW, 1).cuda())
        u64, M64 = u.double().requires_grad_(True), M.double().requires_grad_(True)
        ref = torch.matmul(M64, u64)
        ref.backward(go.double())
        u32, M32 = u.clone().requires_grad_(True), M.clone().requires_grad_(True)
        r32 = torch.matmul(M32, u32)
        r32.backward(go)
        print(f"  B={B} HW={HW}: gu {G.rel_err(ud.grad.cpu().double(), u64.grad):.2e} (torch fp32 {G.rel_err(u32.grad.double(), u64.grad):.2e})"
              f"  gM {G.rel_err(Md.grad.cpu().double(), M64.grad):.2e} (torch fp32 {G.rel_err(M32.grad.double(), M64.grad):.2e})")
else:
    for env in ({}, {"PDE_MIX_NO_SPLIT": "1"}):
        print("split" if not env else "fp32 MFMA", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
