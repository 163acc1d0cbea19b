import cProfile, pstats, time, torch, sys, os, contextlib, io
sys.path.insert(0, os.getcwd())
import cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    l = P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda()
x = torch.randn(128, 3, 32, 32, device="cuda", requires_grad=True); gy = torch.randn_like(x)
for _ in range(5): l(x).backward(gy)
torch.cuda.synchronize()
def loop():
    for _ in range(200):
        y = l(x); y.backward(gy)
    torch.cuda.synchronize()
t0=time.perf_counter(); loop(); print("per fwd+bwd %.1f us" % ((time.perf_counter()-t0)/200*1e6))
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
