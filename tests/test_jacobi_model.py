"""CPU check of the device SVD algorithm (NumPy model, tests/jacobi_model.py) against the
reference's numpy.linalg results stored in the golden fixtures."""
import numpy as np
import pytest

import jacobi_model as jm
from conftest import load_golden

CASES = ["dense_T8_S400", "tree_T12_S2000", "sparse_T10_S257", "lowrank_T9_S700", "edge_T7_S130",
         "c2_slice", "c3_slice"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["full", "sub"])
def test_model_matches_reference(case, mode):
    g = load_golden(case)
    zero = g[f"{mode}_zero_data"]
    cm = g[f"{mode}_cmats"][~zero]
    Q = len(cm)
    sv, sweeps = jm.jacobi_singular_values(cm.reshape(-1, 16, 16), return_sweeps=True)
    sv = sv.reshape(Q, 3, 16)
    ref = g[f"{mode}_svds"][~zero]
    smax = ref.max(axis=(1, 2), keepdims=True)
    assert (np.abs(sv - ref) <= 1e-6 * ref + 1e-12 * smax).all()
    assert sweeps.max() < jm.MAX_SWEEPS
    # numpy.linalg.matrix_rank rule on our values == on LAPACK's values
    np.testing.assert_array_equal(jm.rank_from_sv(sv), jm.rank_from_sv(ref))
    sc, topo, _ = jm.scores_from_sv(sv)
    rsc = g[f"{mode}_rscor"][~zero]
    assert (np.abs(sc - rsc) <= 1e-6 * rsc + 1e-12 * smax[:, 0]).all()
    s = np.sort(rsc, axis=1)
    nondeg = (s[:, 1] - s[:, 0]) > 1e-9 * smax[:, 0, 0]
    np.testing.assert_array_equal(topo[nondeg], g[f"{mode}_rstat"][~zero, 0][nondeg])
