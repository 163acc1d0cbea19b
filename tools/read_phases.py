"""Diagnostic: run the bench workload's backward with a PDE_STAMP build (tools/variant.sh <tag> -DPDE_STAMP=1) and print the
cycle stamps of waves 0 and 4 of one workgroup over one time step (three items), phase by phase."""
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
with torch.no_grad():
    layer.alpha_time_coeff.normal_(0, 0.1); layer.beta_time_coeff.normal_(0, 0.1)
u = torch.randn(512, 64, 32, 32, device="cuda", requires_grad=True)
gy = torch.randn_like(u)
for it in range(3):
    y = layer(u); y.backward(gy)
torch.cuda.synchronize()
ws = keep["ws"]
from cnn_with_pde_amd import _lib as L
lib = L.load()
d = F_._make_desc(512, 64, 32, 0, [s for st in P.adi_schedule(0.001, 1, 1, 10) for s in st], False, 10.0, 1e-6)
total = lib.pde_adi_backward_workspace_bytes(C.byref(d), 0)
off = total - 2048
raw = ws[off:off + 2048].cpu().view(torch.int64).tolist()
st = raw[128:128 + 36]
names = ["start", "relayout/xchg", "solve_adj", "relayout/kap", "state", "end"]
for half in range(2):
    print("wave", 4 * half)
    base = None
    for item in range(3):
        v = st[(half * 3 + item) * 6:(half * 3 + item) * 6 + 6]
        if base is None: base = v[0]
        print("  item", item, "start +%6d" % (v[0] - base), " phases:", " ".join(f"{names[k + 1]}={v[k + 1] - v[k]}" for k in range(5)), " total", v[5] - v[0])
