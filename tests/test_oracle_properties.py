"""Properties of the CPU oracle that the domain offers: linearity in u, constants map
to c/(1+eps)^S, fp64 gradcheck, and the two structural equivalences between reference variants.
These are the same properties the GPU tests use at full size."""
import torch

from oracle import pde_oracle as O


def _params(spec, variant, seed=0, slope=0.0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    p = O.adi_init_params(spec, variant, dtype=dtype, gen=g)
    for k in ("alpha_base", "beta_base"):
        p[k] = p[k] * (1 + 0.2 * torch.randn(p[k].shape, generator=g, dtype=dtype))
    for k in ("alpha_time_coeff", "beta_time_coeff"):
        p[k] = slope * torch.randn(p[k].shape, generator=g, dtype=dtype)
    return p


def test_linearity_and_constants():
    spec = O.cifar10_spec(12, 3, dt=0.05, num_steps=3)
    p = _params(spec, "cifar10", 1, slope=0.5)
    p["channel_mixing"] = torch.eye(3, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 3, 12, 12, generator=g, dtype=torch.float64)
    b = torch.randn(2, 3, 12, 12, generator=g, dtype=torch.float64)
    f = lambda u: O.adi_forward(u, p, spec)
    assert torch.allclose(f(2.5 * a - 0.7 * b), 2.5 * f(a) - 0.7 * f(b), atol=1e-12)
    S = len(O.sweep_schedule(spec))
    c = torch.full((1, 3, 12, 12), 3.0, dtype=torch.float64)
    assert torch.allclose(f(c), c / (1 + spec.eps) ** S, atol=1e-12)      # row sums of A are 1


def test_gradcheck_fp64():
    spec = O.AdiSpec(8, 2, 0.1, 1.0, 1.5, 2, "strang", True, 5.0, "pre", True)
    g = torch.Generator().manual_seed(3)
    p = _params(spec, "cifar10", 3, slope=0.3)
    p["skip_weight"] = torch.tensor(0.2, dtype=torch.float64)
    names = sorted(p)
    u = torch.randn(2, 2, 8, 8, generator=g, dtype=torch.float64, requires_grad=True)
    vals = [p[k].clone().requires_grad_(True) for k in names]

    def f(u_, *ps):
        return O.adi_forward(u_, dict(zip(names, ps)), spec)
    assert torch.autograd.gradcheck(f, (u, *vals), eps=1e-6, atol=1e-6, rtol=1e-5)


def test_svhn_identity_equals_mnist():
    """SVHN layer with coupling = I and skip -> -inf is the mnist layer."""
    g = torch.Generator().manual_seed(4)
    sm = O.mnist_spec(28, 0.01, 1.0, 1.0, 3)
    ss = O.svhn_spec(28, 1, 0.01, 1.0, 3)
    p = _params(sm, "mnist", 4, dtype=torch.float32)
    u = torch.randn(2, 1, 28, 28, generator=g)
    ps = {k: v.unsqueeze(0) for k, v in p.items()}
    ps["channel_coupling"] = torch.eye(1)
    ps["skip_weight"] = torch.tensor(-40.0)
    assert torch.equal(O.adi_forward(u, p, sm), O.adi_forward(u, ps, ss))


def test_tiny_two_forms_agree():
    g = torch.Generator().manual_seed(5)
    p = {"alpha_base": torch.tensor([0.05, 0.3, -1.0]), "channel_scaling": torch.tensor([1.0, 0.8, 1.3])}
    u = torch.randn(2, 3, 16, 16, generator=g)
    a = O.tiny_forward(u, p, dt=0.2, num_steps=2)
    b = O.tiny_forward_faithful(u, p, dt=0.2, num_steps=2)
    assert torch.allclose(a, b, atol=2e-6)
