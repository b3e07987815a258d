"""HIP path end to end vs the reference's golden vectors and the CPU oracle (pytest -m gpu)."""
import pytest

import e2e_checks as ec


@pytest.mark.gpu
@pytest.mark.parametrize("check", ec.ALL_CHECKS, ids=[c.__name__ for c in ec.ALL_CHECKS])
def test_e2e(check):
    res = check()
    bad = [(label, err, tol) for (label, err, tol) in res if not err <= tol]
    assert not bad, "\n".join(f"{l}: err {e:.3e} > tol {t:g}" for l, e, t in bad)
