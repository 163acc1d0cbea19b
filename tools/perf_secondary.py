"""Forward / forward+backward time of BASELINE configs[1] WITH channel mixing (bench `secondary`), for tuning the
per-step path (PDE_MIX_CHUNK etc.)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.EnhancedDiffusionLayer(32, 64, num_steps=10).cuda()
u = torch.randn(512, 64, 32, 32, device="cuda", requires_grad=True); gy = torch.randn_like(u)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwd():
    with torch.no_grad(): layer(u)
def both():
    for p in layer.parameters(): p.grad = None
    u.grad = None
    layer(u).backward(gy)
print("PDE_MIX_CHUNK=%s forward %.3f ms, forward+backward %.3f ms" % (os.environ.get("PDE_MIX_CHUNK", "-"), t(fwd), t(both)))
