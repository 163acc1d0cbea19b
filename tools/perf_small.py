"""The reference's own sizes: the three cifar10 PDE layers (cifar10.py:251-256, C = 3) on a 128-sample
batch, forward + backward — bound by the host's launch rate, not by the device."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layers = [P.EnhancedDiffusionLayer(32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).cuda(),
              P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(),
              P.EnhancedDiffusionLayer(32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).cuda()]
x = torch.randn(128, 3, 32, 32, device="cuda", requires_grad=True)
gy = torch.randn_like(x)
def step():
    for l in layers:
        for p in l.parameters():
            p.grad = None
    x.grad = None
    sum(l(x) for l in layers).backward(gy)
for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("three cifar10 layers (C=3, B=128) fwd+bwd, one call per layer: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
w = torch.softmax(torch.zeros(3, device="cuda"), 0).requires_grad_(True)
def step2():
    for l in layers:
        for p in l.parameters():
            p.grad = None
    x.grad = None
    out, _ = P.diffuse_shared_input(layers, x, w)
    out.backward(gy)
for _ in range(5):
    step2()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    step2()
torch.cuda.synchronize()
print("the same three layers in ONE launch per pass (diffuse_shared_input, weighted sum in the kernel): %.3f ms"
      % ((time.perf_counter() - t0) / 50 * 1e3))
if os.environ.get("PDE_PROFILE_HOST"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        step2()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
