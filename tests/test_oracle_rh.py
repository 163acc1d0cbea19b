"""CPU gate: the oracle's restatement of the Ruthotto-Haber blocks (cifar_2version.py:190-258) against the vectors the
reference's own modules produced (tests/golden_models/model_rh_*.npz).  Tolerance 5e-6 relative (max-norm): the same fp32
arithmetic, but the reference goes through torch's fused batch-norm while the restatement spells it out."""
import pytest

import golden_util as G
import rh_util as R


@pytest.mark.parametrize("name", R.NAMES)
def test_oracle_matches_reference_modules(name):
    g = R.RhGolden(name)
    y, gu, grads, bufs = R.oracle_run(g.oracle_fn(), g.u, g.params, g.bufin, g.gy)
    errs = {"y": G.rel_err(y, g.y), "gu": G.rel_err(gu, g.gu)}
    for n, v in grads.items():
        errs["g_" + n] = G.rel_err(v, g.grads[n])
    for n, v in bufs.items():
        errs["buf_" + n] = G.rel_err(v, g.bufout[n])
    bad = {k: v for k, v in errs.items() if not v <= 5e-6}
    assert not bad, (bad, errs)
