"""Phase-by-phase timing of the bench workload with progress lines (diagnostics)."""
import contextlib, io, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_with_pde_amd as P

def log(*a):
    print(*a, flush=True)

phase = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.EnhancedDiffusionLayer(32, C, num_steps=10, channel_mixing_enabled=(phase == "mix")).to(dev)
u = torch.randn(B, C, 32, 32, device=dev, requires_grad=True)
gy = torch.randn(B, C, 32, 32, device=dev)
log("phase", phase, "B", B, "C", C)
def sync(): torch.cuda.synchronize()
if phase in ("fwd", "fwdbwd", "mix"):
    for it in range(3):
        t0 = time.perf_counter(); y = layer(u); sync(); log("fwd ms", (time.perf_counter() - t0) * 1e3)
        if phase != "fwd":
            t0 = time.perf_counter(); y.backward(gy); sync(); log("bwd ms", (time.perf_counter() - t0) * 1e3)
log("done")
