#!/usr/bin/env python3
"""Diagnostic: the hand-scheduled backward kernel (gen_adi_bwd_asm.py) against the HIP kernel it replaces, on the same
seeded inputs, one child process per variant (the switch is read once per process).
usage: check_asm_bwd.py            -> runs the children and compares
       check_asm_bwd.py child out  -> one variant (PDE_ASM_BWD / PDE_ASM_NW from the environment)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [  # B, C, steps, dt, slope, rel
    (512, 64, 10, 0.001, 0.1, 0.1),
    (37, 5, 3, 0.02, 1.0, 0.15),
    (3, 8, 2, 0.05, 0.5, 0.15),
    (100, 16, 4, 0.01, 0.0, 0.1),
    (1, 1, 1, 0.01, 0.3, 0.1),
]


def child(out):
    import numpy as np
    import torch
    import cnn_with_pde_amd.functional as F
    res = {}
    for ci, (B, C, steps, dt, slope, rel) in enumerate(CASES):
        g = torch.Generator().manual_seed(1000 + ci)
        N = 32
        ab = (2.0 * (1 + rel * torch.randn(C, N, N, generator=g))).cuda().requires_grad_(True)
        bb = (1.8 * (1 + rel * torch.randn(C, N, N, generator=g))).cuda().requires_grad_(True)
        asl = (slope * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True)
        bsl = (slope * torch.randn(C, N, N, generator=g)).cuda().requires_grad_(True)
        u = torch.randn(B, C, N, N, generator=g).cuda().requires_grad_(True)
        gy = torch.randn(B, C, N, N, generator=g).cuda()
        sweeps = [s for st in F.adi_schedule(dt, 1.0, 1.0, steps, "strang") for s in st]
        F.timing_enable(True)
        for rep in range(3 if B >= 512 else 1):
            for t in (ab, bb, asl, bsl, u):
                t.grad = None
            y = F.adi_diffuse(u, ab, bb, asl, bsl, sweeps, checkpoints=0)
            y.backward(gy)
        torch.cuda.synchronize()
        f, nf, b, nb = F.timing_read()
        print(f"case {ci} B={B} C={C} steps={steps}: fwd {f / max(nf, 1) * 1e3:.1f} us  bwd {b / max(nb, 1) * 1e3:.1f} us", flush=True)
        for name, t in (("y", y), ("gu", u.grad), ("gab", ab.grad), ("gbb", bb.grad), ("gas", asl.grad), ("gbs", bsl.grad)):
            t = t.detach().float()
            if t.numel() > (1 << 20):       # big tensors: a few whole samples plus per-sample sums (keeps the file small)
                res[f"{ci}_{name}_sum"] = t.sum(dim=(2, 3)).cpu().numpy()
                t = torch.cat([t[:3], t[t.shape[0] // 2:t.shape[0] // 2 + 2], t[-3:]])
            res[f"{ci}_{name}"] = t.cpu().numpy()
    np.savez(out, **res)


def main():
    import numpy as np
    outs = {}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for tag, env in (("hip", {"PDE_ASM_BWD": "0", "PDE_ASM_FWD": "0"}), ("asm12", {"PDE_ASM_BWD": "1", "PDE_ASM_VARIANT": "12"}),
                     ("asm8b", {"PDE_ASM_BWD": "1", "PDE_ASM_VARIANT": "8b"})):
        out = os.path.join(ROOT, "gpurun_out", f"asmchk_{tag}.npz")
        print("==", tag, flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", out], env=dict(os.environ, **env), timeout=600)
        if r.returncode != 0:
            print(tag, "FAILED rc", r.returncode, flush=True)
            continue
        outs[tag] = dict(np.load(out))
    ref = outs.get("hip")
    ok = True
    for tag in ("asm12", "asm8b"):
        if tag not in outs or ref is None:
            ok = False
            continue
        for k in sorted(ref):
            a, b = outs[tag][k], ref[k]
            den = max(float(np.abs(b).max()), 1e-30)
            err = float(np.abs(a - b).max()) / den
            flag = "" if err <= 2e-6 and np.isfinite(a).all() else "   <-- BAD"
            if flag:
                ok = False
            print(f"{tag} {k:8s} rel {err:.3e} ref max {den:.3e}{flag}", flush=True)
    print("RESULT", "OK" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        sys.exit(main())
