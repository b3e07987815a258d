"""MViTv2 path on the GPU: kernels vs CPU fp32 restatements, encoder vs the reference's golden vectors (pytest -m gpu)."""
import pytest

import mvit_checks as mc


@pytest.mark.gpu
@pytest.mark.parametrize("check", mc.ALL_CHECKS, ids=[c.__name__ for c in mc.ALL_CHECKS])
def test_mvit(check):
    res = check()
    bad = [(label, err, tol) for (label, err, tol) in res if not err <= tol]
    assert not bad, "\n".join(f"{l}: err {e:.3e} > tol {t:g}" for l, e, t in bad)
