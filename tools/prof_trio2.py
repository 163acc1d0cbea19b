"""Diagnostic: the three cifar10 layers (C = 3, batch 128) eager, native host path against the ctypes path, same box."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
from cnn_with_pde_amd import _lib as L
with contextlib.redirect_stdout(io.StringIO()):
    trio = [P.EnhancedDiffusionLayer(32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).cuda(),
            P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(),
            P.EnhancedDiffusionLayer(32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).cuda()]
    one = P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda()
x = torch.randn(128, 3, 32, 32, device="cuda", requires_grad=True); gx = torch.randn_like(x)
w = torch.full((3,), 1.0 / 3, device="cuda", requires_grad=True)
def step():
    out, _ = P.diffuse_shared_input(trio, x, w)
    out.backward(gx)
def step1():
    one(x).backward(gx)
ext = L.host_ext()
for name, fn in (("trio", step), ("one cifar10 layer", step1)):
    for rep in range(2):
        for tag, h in (("native host path", ext), ("ctypes path", False)):
            L._host = h
            for _ in range(50): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(500): fn()
            torch.cuda.synchronize(); print("%s, %s: %.1f us per fwd+bwd" % (name, tag, (time.perf_counter() - t0) / 500 * 1e6))
L._host = ext
