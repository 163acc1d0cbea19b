#!/usr/bin/env python3
"""Diagnostic: the assembly forward kernel with a stop stage (PDE_ASM_FWD_STAGE), lowest first, one child per stage."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
import cnn_with_pde_amd.functional as F
B, C, N, steps = 200, 8, 32, 3
g = torch.Generator().manual_seed(5)
ab = (2.0 * (1 + 0.1 * torch.randn(C, N, N, generator=g))).cuda()
bb = (1.8 * (1 + 0.1 * torch.randn(C, N, N, generator=g))).cuda()
asl = (0.1 * torch.randn(C, N, N, generator=g)).cuda()
bsl = (0.1 * torch.randn(C, N, N, generator=g)).cuda()
u = torch.randn(B, C, N, N, generator=g).cuda()
sweeps = [s for st in F.adi_schedule(0.01, 1.0, 1.0, steps, "strang") for s in st]
with torch.no_grad():
    y = F.adi_diffuse(u, ab, bb, asl, bsl, sweeps, checkpoints=0)
torch.cuda.synchronize()
print("stage ok", flush=True)
''' % ROOT
for stage in (1, 3, 4, 5, 0):
    env = dict(os.environ, PDE_ASM_FWD="1", PDE_ASM_FWD_STAGE=str(stage))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    ok = r.returncode == 0 and "stage ok" in r.stdout
    print(f"fwd stage={stage}: rc={r.returncode} {'OK' if ok else 'FAIL'}", flush=True)
    if not ok:
        print(r.stderr[-800:], flush=True)
        sys.exit(1)
print("ALL STAGES OK")
