"""Channel operator alone at (512, C, 32, 32) fp32: forward (out = M u) and backward (gu = M^T g, gM = g u^T) through
the C ABI, timed with events; the copy floor beside them.  PDECNN_LIB selects a library build.

The event time per call includes host gaps (one autograd call costs the host ~100 us, more than these kernels take):
for KERNEL durations run it under ``rocprofv3 --kernel-trace --stats --output-format csv -d <dir>`` and read the
averages with ``tools/kstat.py <dir>``."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_with_pde_amd import functional as F_
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
W = int(sys.argv[3]) if len(sys.argv) > 3 else 32        # 33, 34, 36: plane strides that are not a power of two
u = torch.randn(512, C, 32, W, device="cuda").to(dt).requires_grad_(True)
M = (torch.eye(C) + 0.05 * torch.randn(C, C)).cuda().requires_grad_(True)
g = torch.randn(512, C, 32, W, device="cuda").to(dt)
def ev(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def fwd():
    with torch.no_grad(): F_.channel_mix(u, M)
y = F_.channel_mix(u, M)
def bwd():
    torch.autograd.grad(y, [u, M], g, retain_graph=True)
v = torch.empty_like(u)
nb = u.numel() * u.element_size()
tf, tb, tc = ev(fwd), ev(bwd), ev(lambda: v.copy_(u))
print("%s C=%d W=%d %s: forward %.1f us (%.2f TB/s), backward %.1f us (%.2f TB/s), copy %.1f us (%.2f TB/s)" % (
    os.path.basename(os.environ.get("PDECNN_LIB", "default")), C, W, str(dt)[6:], tf, 2 * nb / tf / 1e6, tb, 3 * nb / tb / 1e6, tc, 2 * nb / tc / 1e6))
