"""Run one backward of the bench workload with the stamped library and print phase durations."""
import contextlib, io, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
from cnn_with_pde_amd import functional as F_
orig = F_._workspace
keep = {}
def ws_hook(n, dev):
    t = orig(n, dev); keep["ws"] = t; return t
F_._workspace = ws_hook
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.EnhancedDiffusionLayer(32, 64, num_steps=10, channel_mixing_enabled=False).cuda()
u = torch.randn(512, 64, 32, 32, device="cuda", requires_grad=True)
gy = torch.randn_like(u)
for it in range(3):
    y = layer(u); y.backward(gy)
torch.cuda.synchronize()
ws = keep["ws"]     # last workspace allocated = the backward's
from cnn_with_pde_amd import _lib as L
lib = L.load()
# offset of the partials inside the backward workspace: coef + tab + flags
d = F_._make_desc(512, 64, 32, 0, [s for st in P.adi_schedule(0.001, 1, 1, 10) for s in st], False, 10.0, 1e-6)
total = lib.pde_adi_backward_workspace_bytes(C.byref(d), 0)
off = total - 512 - 0          # diagnostics scratch sits right before the (empty) checkpoint area
st = ws[off:off + 48 * 8].cpu().view(torch.int64).tolist()
# timeline of sweeps 20..15 (processing order) for a lower wave (3) and an upper wave (7) of one workgroup
t0 = min(v for v in st if v > 0)
for w, base in (("wave3 (lower)", 0), ("wave7 (upper, one sweep behind)", 24)):
    print(w)
    for i in range(6):
        e, b, d, s_ = (st[base + 4 * i + k] - t0 for k in range(4))
        ax = "x" if (20 - i) % 3 != 1 else "y"
        print(f"  sweep {20 - i} ({ax}): enter {e:6d}  body {b - e:5d}  dma-wait {d - b:4d}  barrier {s_ - d:5d}  -> leaves at {s_:6d}")
