"""Diagnostic: the headline step (512 x 64 x 32 x 32, 10 Strang steps, no channel operator) eager under the three checkpoint
policies — "auto" (the backward waits for this call's coefficient maxima: the host can run at most one forward ahead of the
device), "lagged" (plans from the previous call's maxima, no wait) and a frozen plan (explicit masks) — and replayed from a
hipGraph.  On a quiet host they are the same; a host that is late by more than a forward's duration now and then costs the
"auto" loop that much per step."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.EnhancedDiffusionLayer(32, 64, dt=0.001, num_steps=10, channel_mixing_enabled=False).cuda()
u = torch.randn(512, 64, 32, 32, device="cuda", requires_grad=True); gy = torch.randn_like(u)
def step():
    for p in layer.parameters(): p.grad = None
    u.grad = None
    layer(u).backward(gy)
def run(name, n=100):
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); print("%-8s %.4f ms per step" % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)
for rep in range(2):
    layer.checkpoint_policy = "auto"; run("auto")
    layer.checkpoint_policy = "lagged"; run("lagged")
    layer.freeze_checkpoint_plan(); run("frozen")
