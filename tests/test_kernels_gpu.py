"""HIP kernels vs CPU fp32 restatements, through the C ABI (pytest -m gpu)."""
import pytest

import kernel_checks as kc


@pytest.mark.gpu
@pytest.mark.parametrize("check", kc.ALL_CHECKS, ids=[c.__name__ for c in kc.ALL_CHECKS])
def test_kernel(check):
    bad = [(label, err, tol) for (label, err, tol) in check() if not err <= tol]
    assert not bad, "\n".join(f"{l}: err {e:.3e} > tol {t:g}" for l, e, t in bad)
