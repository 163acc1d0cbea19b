"""Diagnostic: where the host time of the three shared-input cifar10 layers goes — the whole call through
models.diffuse_shared_input against the same launch with every argument prepared once (the extension's `multi` called directly)."""
import contextlib, ctypes as C, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
from cnn_with_pde_amd import _lib as L, functional as F_
with contextlib.redirect_stdout(io.StringIO()):
    trio = [P.EnhancedDiffusionLayer(32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).cuda(),
            P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(),
            P.EnhancedDiffusionLayer(32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).cuda()]
x = torch.randn(128, 3, 32, 32, device="cuda", requires_grad=True); gx = torch.randn_like(x)
w = torch.full((3,), 1.0 / 3, device="cuda", requires_grad=True)
H = L.host_ext()
flat, addrs, keep = [], [], []
for ly in trio:
    st = ly._schedule()
    d = F_._make_desc(128, 3, 32, L.PDE_IO_F32, st.flat, ly._smooth3, ly._clamp_max, ly.stability_eps)
    keep.append(d); addrs.append(C.addressof(d))
    flat += [ly.alpha_base, ly.beta_base, ly.alpha_time_coeff, ly.beta_time_coeff, ly.channel_mixing]
def full():
    out, _ = P.diffuse_shared_input(trio, x, w); out.backward(gx)
def direct():
    res = H.multi(x, w, flat, addrs, 3, False, 1, [], F_.CKPT_AMAX); res[0].backward(gx)
def fwd_only_full():
    with torch.no_grad(): P.diffuse_shared_input(trio, x, w)
def fwd_only_direct():
    with torch.no_grad(): H.multi(x, w, flat, addrs, 3, False, 1, [], F_.CKPT_AMAX)
for name, fn in (("whole call", full), ("prepared arguments", direct), ("forward only, whole call", fwd_only_full), ("forward only, prepared", fwd_only_direct)):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-28s %.1f us to issue, %.1f us per call in all" % (name, (t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6))
