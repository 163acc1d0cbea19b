#!/usr/bin/env python3
"""Diagnostic: cycle stamps of the assembly backward kernel (a build with ABL bit 12, e.g. PDE_ASM_VARIANT=8ba4096): every
wave of workgroup (0,0,0) during time step kk = 5 of its second chunk.  usage: asm_timeline.py <variant>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PDE_ASM_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "8ba4096"
os.environ["PDE_ASM_STAMPS"] = "1"
import torch  # noqa: E402
import cnn_with_pde_amd.functional as F  # noqa: E402
import cnn_with_pde_amd._lib as L  # noqa: E402

B, Cc, N, steps = 512, 64, 32, 10
g = torch.Generator().manual_seed(3)
ab = (2.0 * (1 + 0.1 * torch.randn(Cc, N, N, generator=g))).cuda().requires_grad_(True)
bb = (1.8 * (1 + 0.1 * torch.randn(Cc, N, N, generator=g))).cuda().requires_grad_(True)
asl = (0.1 * torch.randn(Cc, N, N, generator=g)).cuda().requires_grad_(True)
bsl = (0.1 * torch.randn(Cc, N, N, generator=g)).cuda().requires_grad_(True)
u = torch.randn(B, Cc, N, N, generator=g).cuda().requires_grad_(True)
gy = torch.randn(B, Cc, N, N, generator=g).cuda()
sweeps = [s for st in F.adi_schedule(0.001, 1.0, 1.0, steps, "strang") for s in st]
for _ in range(4):
    F.adi_diffuse(u, ab, bb, asl, bsl, sweeps, checkpoints=0).backward(gy)
torch.cuda.synchronize()
lib = L.load()
NM = 16
nw = int("".join(ch for ch in os.environ["PDE_ASM_VARIANT"].split("a")[0] if ch.isdigit()))
buf = (C.c_uint64 * (nw * NM))()
lib.pde_asm_stamps.restype = C.c_int
rc = lib.pde_asm_stamps(buf, nw * NM)
assert rc == 0, rc
names = {0: "step top", 1: "dma issued", 7: "y: adjoint in", 8: "y: H done", 9: "y: G + writes", 10: "y: rows back", 2: "x2 done",
         3: "y done", 4: "x0 done", 5: "drained", 6: "barrier passed"}
order = [0, 1, 2, 7, 8, 9, 10, 3, 4, 5, 6]
t0 = min(buf[w * NM + 0] for w in range(nw) if buf[w * NM + 0])
print("cycles since the first wave's step top; one column per wave")
for m in order:
    row = [buf[w * NM + m] for w in range(nw)]
    print(f"{names[m]:16s} " + " ".join(f"{(x - t0) if x else -1:7d}" for x in row))
print("per-phase deltas (cycles):")
prev = None
for m in order:
    row = [buf[w * NM + m] for w in range(nw)]
    if prev is not None:
        print(f"-> {names[m]:14s} " + " ".join(f"{(x - y) if x and y else -1:7d}" for x, y in zip(row, prev)))
    prev = row
