import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnn_with_pde_amd as P
layer = P.ImprovedDiffusionLayer(64, 64).cuda()
u = torch.randn(256, 64, 64, 64, device="cuda", requires_grad=True); gy = torch.randn_like(u)
for _ in range(8):
    for p in layer.parameters(): p.grad = None
    layer(u).backward(gy)
torch.cuda.synchronize()
