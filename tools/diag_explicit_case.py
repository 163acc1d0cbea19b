"""Diagnostic: one drawn case of tests/test_gpu_fuzz.py::test_random_explicit_case_vs_oracle with bf16 and with fp32 tensors:
the parameter gradients beside the fp32 and fp64 oracle.  usage: diag_explicit_case.py [size C steps B dt]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import golden_util as G
from oracle import pde_oracle as O
import cnn_with_pde_amd as P
a = sys.argv[1:]
case = (int(a[0]), int(a[1]), int(a[2]), int(a[3]), float(a[4]), "bf16") if len(a) >= 5 else (32, 2, 4, 30, 0.5, "bf16")
size, C, steps, B, dt, dtn = case
for dtype in (torch.bfloat16, torch.float32):
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    layer = P.ImprovedDiffusionLayer(size, C, dt=dt, num_steps=steps)
    with torch.no_grad():
        layer.alpha_base.copy_(0.25 * torch.rand(C, generator=g))
        layer.channel_scaling.copy_(1 + 0.3 * torch.randn(C, generator=g))
    u = torch.randn(B, C, size, size, generator=g).to(torch.bfloat16).float()
    gy = torch.randn(B, C, size, size, generator=g).to(torch.bfloat16).float()
    params = {k: v.detach().clone() for k, v in layer.named_parameters() if k != "beta_base"}
    y_ref, gu_ref, gp_ref = O.value_and_grads(lambda a, p: O.tiny_forward(a, p, dt=dt, num_steps=steps), u, params, gy)
    _, _, gp64 = O.value_and_grads(lambda a, p: O.tiny_forward(a, p, dt=dt, num_steps=steps), u.double(), {k: v.double() for k, v in params.items()}, gy.double())
    dl = layer.cuda()
    ud = u.to(dtype).cuda().requires_grad_(True)
    y = dl(ud); y.backward(gy.to(dtype).cuda())
    print(dtype, "alpha", params["alpha_base"].tolist(), "scaling", params["channel_scaling"].tolist())
    print("  g_alpha ours", dl.alpha_base.grad.float().cpu().tolist(), "oracle32", gp_ref["alpha_base"].tolist(), "oracle64", gp64["alpha_base"].tolist())
    print("  g_scaling ours", dl.channel_scaling.grad.float().cpu().tolist(), "oracle32", gp_ref["channel_scaling"].tolist())
    print("  y err", G.rel_err(y.detach().float().cpu(), y_ref), "gu err", G.rel_err(ud.grad.float().cpu(), gu_ref))
