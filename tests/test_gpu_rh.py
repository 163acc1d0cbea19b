"""GPU parity of the Ruthotto-Haber blocks (pde_rh.hip through the C ABI; cifar_2version.py:190-258):
against the vectors the reference's own modules produced, against the pinned oracle on seeded inputs of other sizes, and
the boundary's argument checks.  Tolerance 1e-5 relative (max-norm), fp32."""
import contextlib
import ctypes as C
import io
import json
import os

import numpy as np
import pytest
import torch

import golden_util as G
import rh_util as R

TOL = 1e-5


def _module(g):
    import cnn_with_pde_amd as P
    cls = {"SymmetricLayer": P.SymmetricLayer, "ParabolicBlock": P.ParabolicBlock, "HamiltonianBlock": P.HamiltonianBlock}[g.cls]
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(**g.ctor)
    sd = dict(g.params)
    sd.update(g.bufin)
    m.load_state_dict(sd, strict=True)
    m.train(g.training)
    return m.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", R.NAMES)
def test_blocks_match_reference_vectors(name, monkeypatch):
    g = R.RhGolden(name)
    m = _module(g)
    u = g.u.cuda().requires_grad_(True)
    # the fused kernels must be what runs here (a silent fall-back to plain torch would pass the comparison as well)
    import cnn_with_pde_amd.functional as F_
    calls = []
    orig = F_.sym_layer
    monkeypatch.setattr(F_, "sym_layer", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    y = m(u)
    assert calls, "the module did not go through functional.sym_layer"
    y.backward(g.gy.cuda())
    torch.cuda.synchronize()
    errs = {"y": G.rel_err(y.detach().cpu(), g.y), "gu": G.rel_err(u.grad.cpu(), g.gu)}
    for n, p in m.named_parameters():
        errs["g_" + n] = G.rel_err(p.grad.cpu(), g.grads[n])
    for n, b in m.named_buffers():
        if b.dtype.is_floating_point:
            errs["buf_" + n] = G.rel_err(b.detach().cpu(), g.bufout[n])
        else:
            assert int(b) == int(g.bufout[n]), n                       # num_batches_tracked
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)


@pytest.mark.gpu
def test_symmetric_layer_at_the_reference_size():
    """3 x 32 x 32: K is 3072 x 3072 (rebuilt from its seed — the fixture holds the reference's output, the input and
    BatchNorm gradients and two projections of the 37.7 MB gradient of K)."""
    import cnn_with_pde_amd as P
    z = np.load(os.path.join(R.MODEL_DIR, "model_rh_symmetric_32_train.npz"), allow_pickle=False)
    t = lambda k: torch.from_numpy(np.array(z[k]))
    gen = torch.Generator().manual_seed(95)
    D = 3072
    with contextlib.redirect_stdout(io.StringIO()):
        m = P.SymmetricLayer(3, 32)
    with torch.no_grad():
        m.K.weight.copy_(torch.eye(D) + 0.01 * torch.randn(D, D, generator=gen))
        m.norm.weight.copy_(t("param_norm.weight"))
        m.norm.bias.copy_(t("param_norm.bias"))
    m = m.cuda().train()
    u = t("u").cuda().requires_grad_(True)
    y = m(u)
    y.backward(t("gy").cuda())
    torch.cuda.synchronize()
    gK = m.K.weight.grad.cpu()
    errs = {"y": G.rel_err(y.detach().cpu(), t("y")), "gu": G.rel_err(u.grad.cpu(), t("gu")),
            "g_norm.weight": G.rel_err(m.norm.weight.grad.cpu(), t("grad_norm.weight")),
            "g_norm.bias": G.rel_err(m.norm.bias.grad.cpu(), t("grad_norm.bias")),
            "gK_v1": G.rel_err(gK @ t("v1"), t("gK_v1")), "v2_gK": G.rel_err(t("v2") @ gK, t("v2_gK")),
            "gK_absmax": abs(float(gK.abs().max()) - float(t("gK_absmax"))) / float(t("gK_absmax")),
            "running_mean": G.rel_err(m.norm.running_mean.cpu(), t("bufout_norm.running_mean")),
            "running_var": G.rel_err(m.norm.running_var.cpu(), t("bufout_norm.running_var"))}
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,act,training", [(5, 64, "relu", True), (33, 128, "tanh", True), (130, 192, "relu", True),
                                              (300, 1024, "relu", True), (1, 64, "relu", False), (257, 320, "identity", False),
                                              (512, 128, "relu", True), (700, 192, "tanh", True), (1030, 64, "relu", False)])
def test_symmetric_layer_vs_oracle(B, D, act, training):
    """ragged and single-row batches, every row-block count of the strip kernels, the three activations, both modes
    (no 2-row training batch: BatchNorm over two samples maps every input to +-1, its input gradient is identically zero
    and what both sides return there is rounding noise);
    also the fused residual form (base + scale * ...) against its composition"""
    from cnn_with_pde_amd import functional as F_
    from oracle import pde_oracle as O
    g = torch.Generator().manual_seed(1000 + B + D)
    Kw = torch.eye(D) + 0.05 * torch.randn(D, D, generator=g)
    params = {"K.weight": Kw, "norm.weight": 1 + 0.3 * torch.randn(D, generator=g), "norm.bias": 0.2 * torch.randn(D, generator=g)}
    bufs = {"norm.running_mean": 0.3 * torch.randn(D, generator=g), "norm.running_var": 0.5 + torch.rand(D, generator=g)}
    X = torch.randn(B, D, generator=g)
    base = torch.randn(B, D, generator=g)
    gy = torch.randn(B, D, generator=g)
    scale = 0.7
    fn = lambda u, p: base + scale * (-O.symmetric_layer(u, p, training, act))
    y_ref, gu_ref, gp_ref, buf_ref = R.oracle_run(fn, X, params, bufs, gy)

    bn = torch.nn.BatchNorm1d(D).cuda()
    with torch.no_grad():
        bn.weight.copy_(params["norm.weight"]); bn.bias.copy_(params["norm.bias"])
        bn.running_mean.copy_(bufs["norm.running_mean"]); bn.running_var.copy_(bufs["norm.running_var"])
    bn.train(training)
    Kd = Kw.cuda().requires_grad_(True)
    Xd = X.cuda().requires_grad_(True)
    bd = base.cuda().requires_grad_(True)
    # the module-level policy sends batches above 128 rows (and what BatchNorm1d refuses) to plain torch; the kernels
    # themselves, called here directly, serve every batch
    assert F_.sym_layer_supported(Xd, bn) == (B <= F_.SYM_LAYER_MAX_ROWS)
    y = F_.sym_layer(Xd, Kd, bn, act, base=bd, scale=scale)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    errs = {"y": G.rel_err(y.detach().cpu(), y_ref), "gu": G.rel_err(Xd.grad.cpu(), gu_ref),
            "g_base": G.rel_err(bd.grad.cpu(), gy), "gK": G.rel_err(Kd.grad.cpu(), gp_ref["K.weight"]),
            "g_w": G.rel_err(bn.weight.grad.cpu(), gp_ref["norm.weight"]), "g_b": G.rel_err(bn.bias.grad.cpu(), gp_ref["norm.bias"]),
            "rm": G.rel_err(bn.running_mean.cpu(), buf_ref["norm.running_mean"]),
            "rv": G.rel_err(bn.running_var.cpu(), buf_ref["norm.running_var"])}
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, (bad, errs)


@pytest.mark.gpu
def test_boundary_rejects_what_it_cannot_do():
    from cnn_with_pde_amd import _lib as L
    lib = L.load()
    assert lib.pde_sym_layer_supported(512, 3072) == 1 and lib.pde_sym_layer_supported(5000, 3072) == 1
    assert lib.pde_sym_layer_supported(0, 3072) == 0 and lib.pde_sym_layer_supported(8, 96) == 0
    x = torch.zeros(8, 64, device="cuda")
    k = torch.zeros(64, 64, device="cuda")
    v = torch.zeros(64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok = lambda B, D, act, X: lib.pde_sym_layer_forward(B, D, act, 1, X, p(k), p(v), p(v), None, None, 0.1, 1e-5, None, -1.0,
                                                        p(x), p(x), p(v), p(v), p(x), None, 0, st)
    assert ok(8, 64, 1, p(x)) == 0
    assert ok(8, 64, 1, None) == -1            # null pointer
    assert ok(8, 64, 7, p(x)) == -1            # unknown activation
    assert ok(8, 100, 1, p(x)) == -1           # feature count not a multiple of 64
    # eval mode needs running statistics
    assert lib.pde_sym_layer_forward(8, 64, 1, 0, p(x), p(k), p(v), p(v), None, None, 0.1, 1e-5, None, -1.0,
                                     p(x), p(x), p(v), p(v), p(x), None, 0, st) == -1
    # a workspace that is too small for the split kernels is refused, not silently ignored
    assert lib.pde_sym_layer_workspace_bytes(8, 3072) > 0 and lib.pde_sym_layer_workspace_bytes(300, 3072) == 0
    x3 = torch.zeros(8, 3072, device="cuda"); k3 = torch.zeros(3072, 3072, device="cuda"); v3 = torch.zeros(3072, device="cuda")
    small = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert lib.pde_sym_layer_forward(8, 3072, 1, 1, p(x3), p(k3), p(v3), p(v3), None, None, 0.1, 1e-5, None, -1.0,
                                     p(x3), p(x3), p(v3), p(v3), p(x3), p(small), small.numel(), st) == -5
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_training_batch_of_one_row_raises_like_the_reference():
    """torch.nn.BatchNorm1d refuses a training-mode batch of one row (cifar_2version.py:202 goes through it): the
    counterpart must not silently normalise it with the fused kernels."""
    import cnn_with_pde_amd as P
    with contextlib.redirect_stdout(io.StringIO()):
        m = P.SymmetricLayer(1, 8).cuda().train()
    with pytest.raises(ValueError):
        m(torch.randn(1, 1, 8, 8, device="cuda"))
    m.eval()
    assert m(torch.randn(1, 1, 8, 8, device="cuda")).shape == (1, 1, 8, 8)


@pytest.mark.gpu
def test_split_contraction_under_uneven_load():
    """The 32-column strip kernels leave partial tiles in a scratch buffer that is reused from call to call and added up
    by a second launch (pde_rh.hip): 40 calls with fresh inputs (two batch sizes) while another stream keeps the memory
    system busy, every element of every output compared with plain fp32 torch products (cifar_2version.py:210-219 without
    the module)."""
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(5)
    B, D = 128, 3072
    bn = torch.nn.BatchNorm1d(D).cuda().train()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    worst = 0.0
    for it in range(40):
        Kw = (torch.eye(D) + 0.02 * torch.randn(D, D, generator=g)).cuda().requires_grad_(True)
        X = torch.randn(B if it % 3 else 61, D, generator=g).cuda().requires_grad_(True)
        gy = torch.randn(X.shape[0], D, generator=g).cuda()
        with torch.cuda.stream(side):
            for _ in range(4):
                junk.add_(1.0)
        y = F_.sym_layer(X, Kw, bn, "tanh", base=None, scale=-1.0)
        y.backward(gy)
        Xr = X.detach().clone().requires_grad_(True)
        Kr = Kw.detach().clone().requires_grad_(True)
        P = Xr @ Kr.t()
        Hn = (P - P.mean(0)) / torch.sqrt(P.var(0, unbiased=False) + bn.eps) * bn.weight.detach() + bn.bias.detach()
        yr = -(torch.tanh(Hn) @ Kr)
        yr.backward(gy)
        torch.cuda.synchronize()
        for a, b in ((y, yr), (X.grad, Xr.grad), (Kw.grad, Kr.grad)):
            worst = max(worst, float((a.detach() - b.detach()).abs().max() / b.detach().abs().max()))
        bn.weight.grad = None
        bn.bias.grad = None
    assert worst < 2e-5, worst
