"""GPU parity: BatchNorm2d + the 4x4 average and max pooling behind the feature extractor of cifar10.CIFAR10PDENoConv
(cifar10.py:346-353) as two passes over the activation (pde_bn_pool_*, functional.bn_pool) against torch's own modules
(fp32 reference of the same ops): outputs, input / weight / bias gradients, running statistics; training and eval mode;
and the counterpart model with and without it."""
import contextlib
import copy
import io

import pytest
import torch
import torch.nn as nn

import golden_util as G

pytestmark = pytest.mark.gpu


def _ref(x, bn, gout):
    xr = x.clone().requires_grad_(True)
    f = bn(xr)
    out = torch.cat([nn.functional.adaptive_avg_pool2d(f, 4), nn.functional.adaptive_max_pool2d(f, 4)], dim=1)
    out.backward(gout)
    return out.detach(), xr.grad, bn.weight.grad, bn.bias.grad


@pytest.mark.parametrize("B,C,N", [(8, 3, 32), (5, 4, 16), (3, 2, 64), (16, 3, 28), (1, 3, 32), (128, 3, 32)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_pool_vs_torch(B, C, N, training):
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(100 * B + N + C)
    x = (1.5 * torch.randn(B, C, N, N, generator=g) + 0.3).cuda()
    gout = torch.randn(B, 2 * C, 4, 4, generator=g).cuda()
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        bn.bias.copy_(0.1 * torch.randn(C, generator=g))
        bn.running_mean.copy_(0.2 * torch.randn(C, generator=g))
        bn.running_var.copy_(1 + 0.3 * torch.rand(C, generator=g))
    bn.train(training)
    if B * N * N == 1:
        pytest.skip("batch statistics need more than one value per channel")
    mine = copy.deepcopy(bn)
    ref = _ref(x, bn, gout)
    assert F_.bn_pool_supported(x, mine)
    xm = x.clone().requires_grad_(True)
    out = F_.bn_pool(xm, mine)
    out.backward(gout)
    tol = 2e-5 if training else 1e-6
    assert G.rel_err(out.detach().cpu(), ref[0].cpu()) <= tol
    assert G.rel_err(xm.grad.cpu(), ref[1].cpu()) <= 5 * tol
    assert G.rel_err(mine.weight.grad.cpu(), ref[2].cpu()) <= 5 * tol
    assert G.rel_err(mine.bias.grad.cpu(), ref[3].cpu()) <= 5 * tol
    assert G.rel_err(mine.running_mean.cpu(), bn.running_mean.cpu()) <= 1e-5
    assert G.rel_err(mine.running_var.cpu(), bn.running_var.cpu()) <= 1e-5
    assert int(mine.num_batches_tracked) == int(bn.num_batches_tracked)


def test_bn_pool_unsupported_shapes_are_reported():
    from cnn_with_pde_amd import functional as F_
    bn = nn.BatchNorm2d(3).cuda()
    assert not F_.bn_pool_supported(torch.zeros(2, 3, 30, 30, device="cuda"), bn)          # 30 is not a multiple of 4
    assert not F_.bn_pool_supported(torch.zeros(2, 3, 32, 32, device="cuda", dtype=torch.bfloat16), bn)
    assert not F_.bn_pool_supported(torch.zeros(2, 3, 128, 128, device="cuda"), bn)


def test_counterpart_model_with_and_without_the_fused_tail():
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(8)
    torch.manual_seed(1234)                                    # the model's own initialisation
    with contextlib.redirect_stdout(io.StringIO()):
        model = P.CIFAR10PDENoConv(dropout_rate=0.0).cuda()
    other = copy.deepcopy(model)
    other.fused_tail = False
    x = torch.randn(16, 3, 32, 32, generator=g).cuda()
    tgt = torch.randint(0, 10, (16,), generator=g).cuda()
    res = []
    for m in (model, other):
        m.train()
        loss = nn.functional.cross_entropy(m(x), tgt)
        loss.backward()
        res.append((float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                    m.feature_bn.running_mean.clone(), m.feature_bn.running_var.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-4 * abs(res[1][0])
    # (a Linear bias in front of a BatchNorm1d has gradient zero up to rounding: absolute, not relative, there)
    scale = max(float(t.abs().max()) for t in res[1][1].values())
    errs = {n: float((res[0][1][n] - res[1][1][n]).abs().max()) / max(float(res[1][1][n].abs().max()), 1e-3 * scale)
            for n in res[1][1]}
    # five BatchNorm1d layers on 16 samples sit between the tail and the loss: rounding differences of 1e-6 in the pooled
    # features come back amplified
    bad = {n: e for n, e in errs.items() if not e <= 2e-3}
    assert not bad, bad
    assert G.rel_err(res[0][2].cpu(), res[1][2].cpu()) <= 1e-5 and G.rel_err(res[0][3].cpu(), res[1][3].cpu()) <= 1e-5
