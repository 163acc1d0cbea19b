#!/usr/bin/env python3
"""Diagnostic: launch times (HIP events inside the library) of the cfg2 backward with the HIP kernel and with the
hand-scheduled assembly kernels, one child process per variant.  usage: perf_asm_bwd.py [tag=ENV=VAL,ENV=VAL ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import cnn_with_pde_amd.functional as F
    B, C, N, steps = 512, 64, 32, 10
    g = torch.Generator().manual_seed(3)
    ab = (2.0 * (1 + 0.1 * torch.randn(C, N, N, generator=g))).cuda().requires_grad_(True)
    bb = (1.8 * (1 + 0.1 * torch.randn(C, N, N, generator=g))).cuda().requires_grad_(True)
    asl = (0.1 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True)
    bsl = (0.1 * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True)
    u = torch.randn(B, C, N, N, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(B, C, N, N, generator=g).cuda()
    sweeps = [s for st in F.adi_schedule(0.001, 1.0, 1.0, steps, "strang") for s in st]
    F.timing_enable(True)

    def run(n):
        for _ in range(n):
            for t in (ab, bb, asl, bsl, u):
                t.grad = None
            F.adi_diffuse(u, ab, bb, asl, bsl, sweeps, checkpoints=0).backward(gy)
        torch.cuda.synchronize()
    run(5)
    f0, nf0, b0, nb0 = F.timing_read()
    run(20)
    f1, nf1, b1, nb1 = F.timing_read()
    print(f"fwd {(f1 - f0) / (nf1 - nf0) * 1e3:7.1f} us   bwd {(b1 - b0) / (nb1 - nb0) * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    specs = sys.argv[1:] or ["hip=PDE_ASM_BWD=0,PDE_ASM_FWD=0", "asm8=PDE_ASM_VARIANT=8", "asm12=PDE_ASM_VARIANT=12"]
    for spec in specs:
        tag, _, envs = spec.partition("=")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
        print(f"{tag:12s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED ' + r.stderr[-300:]}", flush=True)
