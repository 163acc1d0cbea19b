"""Diagnostic: the BASELINE configurations with a channel operator between the steps (cfg2 with mixing, cfg3 at 32 channels,
cfg4 bf16 128 channels), forward+backward per call, for the values of PDE_MIX_CHUNK_MB given on the command line
(0 = no batch blocking).  One child process per setting."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import contextlib, io, sys, time, torch
sys.path.insert(0, %r)
import cnn_with_pde_amd as P
def run(name, layer, shape, dtype, steps):
    layer = layer.cuda()
    g = torch.Generator().manual_seed(1)
    u = torch.randn(*shape, generator=g).to(dtype).cuda().requires_grad_(True)
    gy = torch.randn(*shape, generator=g).to(dtype).cuda()
    def fb():
        for p in layer.parameters(): p.grad = None
        u.grad = None
        t0 = time.perf_counter(); y = layer(u); torch.cuda.synchronize(); t1 = time.perf_counter()
        y.backward(gy); torch.cuda.synchronize(); return t1 - t0, time.perf_counter() - t1
    for _ in range(3): fb()
    f = b = 0.0
    for _ in range(steps):
        a_, b_ = fb(); f += a_; b += b_
    print("  %%-10s fwd %%7.3f ms  bwd %%7.3f ms  step %%7.3f ms" %% (name, f / steps * 1e3, b / steps * 1e3, (f + b) / steps * 1e3), flush=True)
with contextlib.redirect_stdout(io.StringIO()):
    c2 = P.EnhancedDiffusionLayer(32, 64, dt=0.001, num_steps=10)
    c32 = P.SvhnDiffusionLayer(28, 32, dt=0.3, num_steps=4)
    c4 = P.SvhnDiffusionLayer(32, 128, num_steps=20)
with torch.no_grad():
    c32.alpha_base.fill_(1.8); c32.beta_base.fill_(1.8); c32.channel_coupling.copy_(torch.eye(32)); c32.skip_weight.fill_(-40.0)
    c4.channel_coupling.copy_(torch.eye(128) + 0.01 * torch.randn(128, 128))
run("cfg2+mix", c2, (512, 64, 32, 32), torch.float32, 10)
run("cfg3_c32", c32, (512, 32, 28, 28), torch.float32, 10)
run("cfg4_bf16", c4, (512, 128, 32, 32), torch.bfloat16, 5)
''' % ROOT
for mb in (sys.argv[1:] or ["0", "32"]):
    print("PDE_MIX_CHUNK_MB=" + mb, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PDE_MIX_CHUNK_MB=mb))
