"""Diagnostic: device time of one SymmetricLayer(3, 32) forward + backward (fused kernels), by CUDA events over 50 calls,
and the worst error against plain torch products over those calls (inputs change every call)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P
from cnn_with_pde_amd import functional as F_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D = 3072
bn = torch.nn.BatchNorm1d(D).cuda().train()
g = torch.Generator().manual_seed(1)
K = (torch.eye(D) + 0.02 * torch.randn(D, D, generator=g)).cuda().requires_grad_(True)
Xs = [torch.randn(B, D, generator=g).cuda().requires_grad_(True) for _ in range(4)]
gy = torch.randn(B, D, generator=g).cuda()
def one(X):
    K.grad = None; X.grad = None
    y = F_.sym_layer(X, K, bn, "relu"); y.backward(gy); return y
for i in range(5): one(Xs[i % 4])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50): one(Xs[i % 4])
e1.record(); torch.cuda.synchronize()
print("B=%d: %.1f us per forward + backward" % (B, e0.elapsed_time(e1) / 50 * 1e3))
worst = 0.0
for i in range(20):
    X = Xs[i % 4]
    y = one(X)
    Pm = X.detach() @ K.detach().t()
    Hn = torch.relu((Pm - Pm.mean(0)) / torch.sqrt(Pm.var(0, unbiased=False) + bn.eps) * bn.weight.detach() + bn.bias.detach())
    yr = -(Hn @ K.detach())
    worst = max(worst, float((y.detach() - yr).abs().max() / yr.abs().max()))
print("worst relative error of the forward against torch: %.2e" % worst)
