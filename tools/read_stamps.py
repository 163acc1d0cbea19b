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
# every wave of one workgroup at the end of barrier intervals 2 and 3: work done / DMA landed / barrier passed
t0 = min(v for v in st if v > 0)
for t in range(2):
    print("interval", 2 + t)
    for w in range(8):
        a_, b_, c_ = (st[(w * 2 + t) * 3 + k] - t0 for k in range(3))
        print(f"  wave {w}: work done {a_:6d}  dma landed +{b_ - a_:4d}  barrier passed {c_:6d} (waited {c_ - b_:5d})")
