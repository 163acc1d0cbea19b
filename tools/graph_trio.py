"""The reference-shape trio (three cifar10 C=3 layers, 128x3x32x32) forward+backward under hipGraph replay
(torch.cuda.CUDAGraph): with an explicit checkpoint mask the library issues launches only (no host wait, no copy), so
a captured step replays without the Python/ctypes path."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layers = [P.EnhancedDiffusionLayer(32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).cuda(),
              P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(),
              P.EnhancedDiffusionLayer(32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).cuda()]
for ly in layers:
    ly.checkpoint_policy = 0          # coefficients of 1e-3: no state needs parking (plan_checkpoints would say the same)
x = torch.randn(128, 3, 32, 32, device="cuda", requires_grad=True)
gy = torch.randn_like(x)
params = [p for ly in layers for p in ly.parameters()]

w = torch.full((3,), 1.0 / 3, device="cuda")
FUSED = len(sys.argv) > 1 and sys.argv[1] == "fused"

def step():
    if FUSED:
        out, _ = P.diffuse_shared_input(layers, x, w)       # the three layers in one launch per pass
    else:
        ys = [ly(x) for ly in layers]
        out = (ys[0] + ys[1] + ys[2]) / 3
    return torch.autograd.grad(out, [x] + params, gy)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    grads = step()
eager = [t.clone() for t in step()]
g.replay(); torch.cuda.synchronize()
err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(grads, eager))
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): g.replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 100 * 1e3
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 100 * 1e3
print(("trio fwd+bwd (%s, checkpoint mask 0):" % ("one launch per pass" if FUSED else "one call per layer")) + " eager %.3f ms, hipGraph replay %.3f ms, max rel diff %.1e" % (t_eager, t_graph, err))
