"""CPU check of the dqds model (tests/dqds_model.py): an alternative to the implicit-QR kernel that
was evaluated and not adopted (negative results).  Kept pinned so that the
comparison can be repeated: singular values and numpy's rank rule on count matrices of the kind the
hot path produces, including rank-deficient ones."""
import numpy as np
import pytest

import dqds_model as dm
from tetrad_amd import synth


def _count_like(rng, n, sparse):
    """Random 16x16 integer matrices shaped like site-pattern counts (a few big cells, many small)."""
    lam = rng.gamma(0.3, 40.0, size=(n, 16, 16))
    if sparse:
        lam *= rng.random((n, 16, 16)) < 0.25
        lam[:, :, rng.integers(0, 16, size=4)] = 0.0          # exact zero columns
    return rng.poisson(lam).astype(np.float64)


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("shift", ["newton", "laguerre"])
def test_dqds_matches_numpy(sparse, shift):
    rng = np.random.default_rng(11 + sparse)
    mats = _count_like(rng, 300, sparse)
    ref = np.linalg.svd(mats, compute_uv=False)
    sv, sweeps = dm.singular_values(mats, shift=shift, return_sweeps=True)
    sv = -np.sort(-sv, axis=1)
    smax = np.maximum(ref[:, :1], 1e-300)
    assert np.max(np.abs(sv - ref) / smax) < 1e-13
    rk_ref = (ref > smax * 16 * dm.EPS).sum(axis=1)
    rk = (sv > sv[:, :1] * 16 * dm.EPS).sum(axis=1)
    np.testing.assert_array_equal(rk, rk_ref)
    assert sweeps.max() < dm.MAX_SWEEPS


def test_bidiagonalize_preserves_singular_values():
    rng = np.random.default_rng(5)
    mats = _count_like(rng, 100, True)
    d, e = dm.bidiagonalize(mats)
    B = np.zeros_like(mats)
    idx = np.arange(16)
    B[:, idx, idx] = d
    B[:, idx[:-1], idx[1:]] = e
    a = np.linalg.svd(mats, compute_uv=False)
    b = np.linalg.svd(B, compute_uv=False)
    assert np.max(np.abs(a - b) / np.maximum(a[:, :1], 1e-300)) < 1e-13
