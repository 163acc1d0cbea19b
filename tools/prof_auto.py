"""Diagnostic: where the host time of the "auto" checkpoint policy goes on a launch-bound shape (mnist layer, 64x1x28x28)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
from cnn_with_pde_amd import functional as F_
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
F_._kmax_channel = timed("kmax_channel", F_._kmax_channel)
F_._KmaxTicket.wait = timed("ticket.wait", F_._KmaxTicket.wait)
F_.plan_checkpoints = timed("plan", F_.plan_checkpoints)
with contextlib.redirect_stdout(io.StringIO()):
    l = P.MnistDiffusionLayer().cuda()
x = torch.randn(64, 1, 28, 28, device="cuda", requires_grad=True); gy = torch.randn_like(x)
for pol in ("auto", 0, "auto"):
    l.checkpoint_policy = pol
    for _ in range(20): l(x).backward(gy)
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter()
    for _ in range(300): l(x).backward(gy)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 300 * 1e6
    print(f"policy {pol}: {tot:.1f} us per fwd+bwd;", {k: round(v / 300 * 1e6, 1) for k, v in acc.items()})
