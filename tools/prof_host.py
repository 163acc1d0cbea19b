"""Host-side profile of one small layer's forward+backward (launch-bound shapes): cProfile of the main thread."""
import cProfile, pstats, time, torch, sys, os, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_with_pde_amd as P
which = sys.argv[1] if len(sys.argv) > 1 else "mnist"
with contextlib.redirect_stdout(io.StringIO()):
    if which == "mnist":
        l, shape = P.MnistDiffusionLayer().cuda(), (64, 1, 28, 28)
    else:
        l, shape = P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).cuda(), (128, 3, 32, 32)
x = torch.randn(*shape, device="cuda", requires_grad=True); gy = torch.randn_like(x)
for _ in range(5): l(x).backward(gy)
torch.cuda.synchronize()
def loop(n=200):
    for _ in range(n):
        y = l(x); y.backward(gy)
    torch.cuda.synchronize()
t0 = time.perf_counter(); loop(); print("%s: %.1f us per fwd+bwd" % (which, (time.perf_counter() - t0) / 200 * 1e6))
l.checkpoint_policy = 0
t0 = time.perf_counter(); loop(); print("%s with checkpoint_policy=0 (no coefficient maxima, no wait): %.1f us" % (which, (time.perf_counter() - t0) / 200 * 1e6))
l.checkpoint_policy = "auto"
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
