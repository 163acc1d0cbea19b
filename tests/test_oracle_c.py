"""Pin the C restatement (plain Thomas, hand-derived gradients) against the torch oracle (which is
itself pinned bitwise against the reference's vectors): an independent check of the adjoint
formulas of SURVEY.md Appendix A.3 that the HIP kernels implement."""
import pytest
import torch

import golden_util as G
from oracle import c_oracle as CO
from oracle import pde_oracle as O

pytestmark = pytest.mark.skipif(not CO.available(), reason="oracle/libpde_oracle_c.so not built (run build())")


@pytest.mark.parametrize("smooth,cmax,split,slope", [(True, None, "strang", 0.0), (False, 10.0, "strang", 2.0),
                                                     (False, 10.0, "lie", 1.0), (True, 3.0, "strang", 4.0)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_c_oracle_matches_torch_oracle(smooth, cmax, split, slope, dtype, tol):
    g = torch.Generator().manual_seed(17)
    spec = O.AdiSpec(12, 3, 0.08, 1.0, 1.4, 3, split, smooth, cmax, "none", False)
    p = {"alpha_base": 1.0 + 0.6 * torch.randn(3, 12, 12, generator=g, dtype=dtype),   # some below eps: clamped
         "beta_base": 2.5 + 1.0 * torch.randn(3, 12, 12, generator=g, dtype=dtype),    # some above 3.0
         "alpha_time_coeff": slope * torch.randn(3, 12, 12, generator=g, dtype=dtype),
         "beta_time_coeff": slope * torch.randn(3, 12, 12, generator=g, dtype=dtype)}
    u = torch.randn(4, 3, 12, 12, generator=g, dtype=dtype)
    gy = torch.randn(4, 3, 12, 12, generator=g, dtype=dtype)
    y0, gu0, gp0 = O.value_and_grads(lambda a, q: O.adi_forward(a, q, spec), u, p, gy)
    y1, gu1, gp1 = CO.adi_value_and_grads(u, p, gy, spec)
    assert G.rel_err(y1, y0) <= tol and G.rel_err(gu1, gu0) <= tol
    for k in gp1:
        assert G.rel_err(gp1[k], gp0[k]) <= tol, k


def test_c_oracle_matches_reference_vectors():
    g = G.Golden("fashion_default_f64")
    spec = g.adi_spec()
    y, gu, gp = CO.adi_value_and_grads(g.u, g.params, g.gy, spec)
    assert G.rel_err(y, g.y) <= 1e-11 and G.rel_err(gu, g.gu) <= 1e-11
    for k in gp:
        assert G.rel_err(gp[k], g.grads[k]) <= 1e-10, k
