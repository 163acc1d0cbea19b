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
from cnn_with_pde_amd import _lib as _L
_ext = _L.host_ext()
for rep in range(2):
    for name, h in (("native host path", _ext), ("ctypes path", False)):
        _L._host = h if h is not None else False
        loop(50)
        t0 = time.perf_counter(); loop(1000); print("%s, %s: %.1f us per fwd+bwd" % (which, name, (time.perf_counter() - t0) / 1000 * 1e6))
_L._host = _ext if _ext is not None else False
l.checkpoint_policy = 0
t0 = time.perf_counter(); loop(); print("%s with checkpoint_policy=0 (no coefficient maxima, no wait): %.1f us" % (which, (time.perf_counter() - t0) / 200 * 1e6))
l.checkpoint_policy = "auto"
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

# the floor of this host: a custom autograd function that launches nothing and allocates one tensor per pass
class _Null(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, a, b, c, d):
        ctx.save_for_backward(u, a, b, c, d)
        return torch.empty_like(u)
    @staticmethod
    def backward(ctx, g):
        u, a, b, c, d = ctx.saved_tensors
        return torch.empty_like(g), torch.empty_like(a), torch.empty_like(b), torch.empty_like(c), torch.empty_like(d)
ps = [torch.zeros(shape[1], shape[2], shape[3], device="cuda", requires_grad=True) for _ in range(4)]
def floor(n=2000):
    for _ in range(n):
        _Null.apply(x, *ps).backward(gy)
    torch.cuda.synchronize()
floor(200)
t0 = time.perf_counter(); floor(); print("null autograd function (1+5 allocations): %.1f us per fwd+bwd" % ((time.perf_counter() - t0) / 2000 * 1e6))
def allocs(n=20000):
    for _ in range(n):
        torch.empty_like(x)
t0 = time.perf_counter(); allocs(); print("torch.empty_like: %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(1000): l(x)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("forward only, no_grad, host time to issue: %.1f us" % ((t1 - t0) / 1000 * 1e6))
