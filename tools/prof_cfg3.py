"""cfg3 (fashion semantics x 32 channels through the SVHN layer) under rocprofv3: python tools/prof_cfg3.py"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.SvhnDiffusionLayer(28, 32, dt=0.3, num_steps=4).cuda()
with torch.no_grad():
    layer.channel_coupling.copy_(torch.eye(32)); layer.skip_weight.fill_(-40.0)
u = torch.randn(512, 32, 28, 28, device="cuda", requires_grad=True); gy = torch.randn_like(u)
for _ in range(5):
    for p in layer.parameters(): p.grad = None
    u.grad = None
    layer(u).backward(gy)
torch.cuda.synchronize()
