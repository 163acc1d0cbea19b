"""The native host path (csrc/host_ext.cpp, a torch C++ autograd node over the C ABI) against the ctypes path
(functional._AdiFn): the same launches with the same arguments, so every output must be bitwise identical — forward,
input gradient and the four coefficient gradients (mnist_test.py:44-198 / fashion_mnist.py:18-196 through autograd)."""
import contextlib
import io

import pytest
import torch


def _layer(kind, dev="cuda"):
    import cnn_with_pde_amd as P
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == "mnist":
            l, shape = P.MnistDiffusionLayer(), (64, 1, 28, 28)
        elif kind == "fashion":                                   # large coefficients: the plan has checkpoints
            l, shape = P.FashionDiffusionLayer(), (37, 1, 28, 28)
        else:
            raise ValueError(kind)
    g = torch.Generator().manual_seed(len(kind))
    with torch.no_grad():
        for p in (l.alpha_base, l.beta_base):
            p.mul_(1 + 0.2 * torch.rand(p.shape, generator=g))
        for p in (l.alpha_time_coeff, l.beta_time_coeff):
            p.copy_(0.5 * torch.randn(p.shape, generator=g))
    return l.to(dev), shape, g


def _run(F_, l, x, gy, sweeps, ck, kw, native):
    from cnn_with_pde_amd import _lib as L
    args = (l.alpha_base, l.beta_base, l.alpha_time_coeff, l.beta_time_coeff)
    for p in args:
        p.grad = None
    xx = x.clone().requires_grad_(True)
    if native:
        assert L.host_ext() is not None
        y = F_.adi_diffuse(xx, *args, sweeps, checkpoints=ck, **kw)
        assert type(y.grad_fn).__name__ != "_AdiFnBackward", "the call did not take the native host path"
    else:
        y = F_._AdiFn.apply(xx, *args, tuple(sweeps), bool(kw["smooth3"]), kw["clamp_max"], float(kw["eps"]), ck, None)
    y.backward(gy)
    torch.cuda.synchronize()
    return [y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in args]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mnist", "fashion"])
@pytest.mark.parametrize("ck", ["auto", 0, 0b101])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_native_host_path_matches_ctypes_path_bitwise(kind, ck, dtype):
    from cnn_with_pde_amd import functional as F_
    l, shape, g = _layer(kind)
    x = torch.randn(*shape, generator=g).to("cuda", dtype)
    gy = torch.randn(*shape, generator=g).to("cuda", dtype)
    sweeps = l._schedule().flat
    kw = dict(smooth3=l._smooth3, clamp_max=l._clamp_max, eps=l.stability_eps)
    a = _run(F_, l, x, gy, sweeps, ck, kw, native=True)
    b = _run(F_, l, x, gy, sweeps, ck, kw, native=False)
    for i, (s, t) in enumerate(zip(a, b)):
        assert s.dtype == t.dtype and s.shape == t.shape, i
        assert torch.equal(s, t), (i, float((s.float() - t.float()).abs().max()))


@pytest.mark.gpu
def test_modules_take_the_native_host_path_and_more_calls_than_slots_may_be_outstanding():
    """300 forward calls whose backward has not run (more than the 256 pinned slots of the ring), then every backward."""
    from cnn_with_pde_amd import _lib as L
    assert L.host_ext() is not None
    l, shape, g = _layer("mnist")
    x = torch.randn(4, 1, 28, 28, generator=g).cuda().requires_grad_(True)
    ys = [l(x) for _ in range(300)]
    assert "AdiFn" in ys[0].grad_fn.name() and type(ys[0].grad_fn).__name__ == "CppFunction"
    torch.stack(ys).sum().backward()
    torch.cuda.synchronize()
    one = l(x.detach().requires_grad_(True))
    gref = torch.autograd.grad(one.sum(), l.alpha_base)[0]
    assert torch.allclose(l.alpha_base.grad, 300 * gref, rtol=1e-4, atol=1e-6)
    # no_grad and inference: nothing is kept, the output is the same
    with torch.no_grad():
        assert torch.equal(l(x), ys[0].detach())


@pytest.mark.gpu
def test_in_place_change_of_a_parameter_between_forward_and_backward_raises():
    l, shape, g = _layer("mnist")
    x = torch.randn(4, 1, 28, 28, generator=g).cuda().requires_grad_(True)
    y = l(x)
    with torch.no_grad():
        l.alpha_base.add_(0.1)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()


def _grads_of(mods, extra=()):
    out = []
    for m in mods:
        out += [p.grad.clone() if p.grad is not None else None for p in m.parameters()]
    return out + [t.grad.clone() if t.grad is not None else None for t in extra]


def _both_paths(fn):
    """fn() once through the native host path and once through the ctypes path (functional._Adi*Fn)."""
    from cnn_with_pde_amd import _lib as L
    ext = L.host_ext()
    assert ext is not None
    a = fn(True)
    L._host = False
    try:
        b = fn(False)
    finally:
        L._host = ext
    assert len(a) == len(b)
    for i, (s, t) in enumerate(zip(a, b)):
        assert (s is None) == (t is None), i
        if s is not None:
            assert s.dtype == t.dtype and s.shape == t.shape, i
            assert torch.equal(s, t), (i, float((s.float() - t.float()).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cifar10", "svhn"])
@pytest.mark.parametrize("ck", ["auto", 0, 0b11])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_one_launch_layers_match_bitwise(which, ck, dtype):
    """cifar10.py:84-112 (operator before every step) and SVHN.py:55-76 (after every step, then the skip blend) at C = 3."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        l = (P.EnhancedDiffusionLayer(32, 3, dt=0.05, num_steps=4) if which == "cifar10"
             else P.SvhnDiffusionLayer(32, 3, dt=0.05, num_steps=3)).cuda()
    with torch.no_grad():
        l.alpha_time_coeff.copy_(torch.randn(l.alpha_time_coeff.shape, generator=g))
        l.beta_base.mul_(3.0)
    l.checkpoint_policy = ck
    x = torch.randn(9, 3, 32, 32, generator=g).to("cuda", dtype)
    gy = torch.randn(9, 3, 32, 32, generator=g).to("cuda", dtype)

    def fn(native):
        for p in l.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        y = l(xx)
        assert (type(y.grad_fn).__name__ == "CppFunction") == native
        y.backward(gy)
        torch.cuda.synchronize()
        return [y.detach().clone()] + _grads_of([l], [xx])
    _both_paths(fn)


@pytest.mark.gpu
@pytest.mark.parametrize("ck", ["auto", 0b01])
@pytest.mark.parametrize("sums,use_w", [(False, True), (True, True), (True, False)])
def test_shared_input_layers_match_bitwise(ck, sums, use_w):
    """cifar10.py:272-280: three layers on one x, softmax-weighted sum, plane sums for the attention pool."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(12)
    with contextlib.redirect_stdout(io.StringIO()):
        trio = [P.EnhancedDiffusionLayer(32, 3, dt=0.01, num_steps=5, dx=1.0, dy=1.0).cuda(),
                P.EnhancedDiffusionLayer(32, 3, dt=0.2, num_steps=8, dx=2.0, dy=2.0).cuda(),
                P.EnhancedDiffusionLayer(32, 3, dt=0.05, num_steps=4, dx=1.5, dy=1.5).cuda()]
    for ly in trio:
        ly.checkpoint_policy = ck
    x = torch.randn(17, 3, 32, 32, generator=g).cuda()
    w = torch.tensor([0.2, 0.5, 0.3]).cuda().requires_grad_(True) if use_w else None
    gq = [torch.randn(17, 3, 32, 32, generator=g).cuda() for _ in range(4)]
    gs = [torch.randn(17, 3, generator=g).cuda() for _ in range(3)]

    def fn(native):
        for ly in trio:
            for p in ly.parameters():
                p.grad = None
        if w is not None:
            w.grad = None
        xx = x.clone().requires_grad_(True)
        res = P.diffuse_shared_input(trio, xx, w, plane_sums=sums)
        out, ys = res[0], res[1]
        assert (type(ys[0].grad_fn).__name__ == "CppFunction") == native
        loss = (ys[0] * gq[1]).sum() + (ys[2] * gq[3]).sum()            # the second layer's output only through `out`
        if out is not None:
            loss = loss + (out * gq[0]).sum()
        if sums:
            loss = loss + (res[2][1] * gs[1]).sum()
        loss.backward()
        torch.cuda.synchronize()
        vals = [t.detach().clone() for t in ys] + ([out.detach().clone()] if out is not None else [])
        if sums:
            vals += [t.detach().clone() for t in res[2]]
        return vals + _grads_of(trio, [xx] + ([w] if w is not None else []))
    _both_paths(fn)


@pytest.mark.gpu
@pytest.mark.parametrize("B,training,with_base", [(64, True, False), (128, True, True), (33, False, True), (300, True, False)])
def test_symmetric_layer_node_matches_bitwise(B, training, with_base):
    """cifar_2version.py:190-258: the C++ node of functional.sym_layer against functional._SymLayerFn (running statistics
    included: both start from the same buffers)."""
    from cnn_with_pde_amd import functional as F_
    g = torch.Generator().manual_seed(B)
    D = 192
    K = (torch.eye(D) + 0.05 * torch.randn(D, D, generator=g)).cuda()
    X = torch.randn(B, D, generator=g).cuda()
    base = torch.randn(B, D, generator=g).cuda() if with_base else None
    gy = torch.randn(B, D, generator=g).cuda()
    rm0, rv0 = 0.1 * torch.randn(D, generator=g).cuda(), (0.5 + torch.rand(D, generator=g)).cuda()

    def fn(native):
        bn = torch.nn.BatchNorm1d(D).cuda().train(training)
        with torch.no_grad():
            bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
        Kp, Xp = K.clone().requires_grad_(True), X.clone().requires_grad_(True)
        bp = None if base is None else base.clone().requires_grad_(True)
        y = F_.sym_layer(Xp, Kp, bn, "tanh", base=bp, scale=0.7)
        node = y.grad_fn.next_functions[0][0]             # behind the view back to X's shape
        assert (type(node).__name__ == "CppFunction") == native and (type(node).__name__ == "_SymLayerFnBackward") != native
        y.backward(gy)
        torch.cuda.synchronize()
        return [y.detach().clone(), Xp.grad, Kp.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(),
                None if bp is None else bp.grad]
    _both_paths(fn)


@pytest.mark.gpu
@pytest.mark.parametrize("which,C_", [("cifar10", 8), ("svhn", 8), ("cifar10", 32)])
@pytest.mark.parametrize("ck", ["auto", 0b10])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_per_step_layers_match_bitwise(which, C_, ck, dtype):
    """Layers with a channel operator at widths above 4 (cifar10.py:84-112, SVHN.py:55-76): per-step launches or, at
    C = 32 fp32, the one-launch forward — the C++ node against functional._AdiMixedFn."""
    import cnn_with_pde_amd as P
    g = torch.Generator().manual_seed(13 + C_)
    with contextlib.redirect_stdout(io.StringIO()):
        l = (P.EnhancedDiffusionLayer(32, C_, dt=0.05, num_steps=3) if which == "cifar10"
             else P.SvhnDiffusionLayer(32, C_, dt=0.05, num_steps=3)).cuda()
    with torch.no_grad():
        l.beta_time_coeff.copy_(torch.randn(l.beta_time_coeff.shape, generator=g))
        l.alpha_base.mul_(2.0)
    l.checkpoint_policy = ck
    x = torch.randn(6, C_, 32, 32, generator=g).to("cuda", dtype)
    gy = torch.randn(6, C_, 32, 32, generator=g).to("cuda", dtype)

    def fn(native):
        for p in l.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        y = l(xx)
        y.backward(gy)
        torch.cuda.synchronize()
        return [y.detach().clone()] + _grads_of([l], [xx])
    _both_paths(fn)


@pytest.mark.gpu
def test_lagged_policy_through_the_native_path():
    """layers.py "lagged": an explicit mask now, this call's coefficient maxima for the next plan.  The native node hands
    them over in a ticket with the ctypes tickets' interface (event.query(), host); three calls (cold cache, warm cache,
    after a parameter change) give bitwise what the ctypes path gives, and the plan follows the coefficients."""
    from cnn_with_pde_amd import _lib as L
    ext = L.host_ext()
    assert ext is not None
    outs = {}
    for native in (True, False):
        L._host = ext if native else False
        try:
            l, shape, g = _layer("fashion")                   # large coefficients: the plans have checkpoints
            l.checkpoint_policy = "lagged"
            x = torch.randn(*shape, generator=g).cuda()
            gy = torch.randn(*shape, generator=g).cuda()
            res = []
            for it in range(3):
                if it == 2:
                    with torch.no_grad():
                        l.alpha_base.mul_(0.01)               # tiny coefficients: the next plan needs no checkpoint
                for p in l.parameters():
                    p.grad = None
                xx = x.clone().requires_grad_(True)
                y = l(xx)
                assert (type(y.grad_fn).__name__ == "CppFunction") == native
                y.backward(gy)
                torch.cuda.synchronize()
                res += [y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in l.parameters()]
            cache = l.__dict__["_kmax_cache"]
            tk, mask = next(iter(cache.values()))
            assert tk.event.query() and len(tk.host.tolist()) == len(l._schedule().flat)
            res.append(torch.tensor(float(mask)))
            outs[native] = res
        finally:
            L._host = ext
    for i, (s, t) in enumerate(zip(outs[True], outs[False])):
        assert torch.equal(s, t), i
