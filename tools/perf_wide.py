"""Forward time of the one-launch C = 32 / 64 path (pde_adi_wide.h) at BASELINE configs[1] with channel mixing.
PDECNN_LIB selects the library build (ablation variants)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.EnhancedDiffusionLayer(32, C, num_steps=10).cuda()
u = torch.randn(512, C, 32, 32, device="cuda")
def fwd():
    with torch.no_grad(): layer(u)
for _ in range(3): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fwd()
torch.cuda.synchronize()
print("%s C=%d forward %.3f ms" % (os.path.basename(os.environ.get("PDECNN_LIB", "default")), C, (time.perf_counter() - t0) / 10 * 1e3))
