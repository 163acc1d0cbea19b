"""Diagnostic: host-side profile of the three cifar10 layers (C = 3, batch 128) in one launch per pass, eager."""
import cProfile, pstats, contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    trio = [P.EnhancedDiffusionLayer(32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).cuda(),
            P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(),
            P.EnhancedDiffusionLayer(32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).cuda()]
x = torch.randn(128, 3, 32, 32, device="cuda", requires_grad=True); gx = torch.randn_like(x)
w = torch.full((3,), 1.0 / 3, device="cuda", requires_grad=True)
def step():
    for ly in trio:
        for p in ly.parameters(): p.grad = None
    x.grad = None
    out, _ = P.diffuse_shared_input(trio, x, w)
    out.backward(gx)
for _ in range(200): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); print("%.1f us per step" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
