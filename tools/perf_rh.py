"""Diagnostic: one SymmetricLayer application (3 x 32 x 32, batch 128, training mode), forward and backward, fused kernels
against the same module in plain torch (rocBLAS); run under rocprofv3 --kernel-trace --stats for the per-kernel times."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with contextlib.redirect_stdout(io.StringIO()):
    sl = P.SymmetricLayer(3, 32).cuda().train()
x = torch.randn(B, 3, 32, 32, device="cuda", requires_grad=True)
g = torch.randn_like(x)
for fused in (True, False):
    sl.fused = fused
    def one():
        for p in sl.parameters(): p.grad = None
        x.grad = None
        t0 = time.perf_counter(); y = sl(x); torch.cuda.synchronize(); t1 = time.perf_counter()
        y.backward(g); torch.cuda.synchronize(); return t1 - t0, time.perf_counter() - t1
    for _ in range(5): one()
    f = b = 0.0
    for _ in range(20):
        a_, b_ = one(); f += a_; b += b_
    print(f"B={B} fused={fused}: fwd {f / 20 * 1e3:.3f} ms  bwd {b / 20 * 1e3:.3f} ms", flush=True)
