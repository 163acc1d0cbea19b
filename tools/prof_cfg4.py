"""cfg4 (SVHN layer, 128 channels, 20 steps, bf16 I/O) under rocprofv3: python tools/prof_cfg4.py"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.SvhnDiffusionLayer(32, 128, num_steps=20).cuda()
u = torch.randn(512, 128, 32, 32, device="cuda").bfloat16().requires_grad_(True); gy = torch.randn_like(u)
for _ in range(3):
    for p in layer.parameters(): p.grad = None
    u.grad = None
    layer(u).backward(gy)
torch.cuda.synchronize()
