import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
with contextlib.redirect_stdout(io.StringIO()):
    layer = P.EnhancedDiffusionLayer(32, 64, num_steps=10).cuda()
u = torch.randn(512, 64, 32, 32, device="cuda", requires_grad=True); gy = torch.randn_like(u)
for _ in range(6):
    for p in layer.parameters(): p.grad = None
    layer(u).backward(gy)
torch.cuda.synchronize()
