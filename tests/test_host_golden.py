"""The host functions either side of the hot path against outputs of the REFERENCE's own
code: tests/golden/host_f2f3.npz, made by tests/golden/make_golden_host.py from
/root/reference/tetrad/src/combinations.py (imported as is) and from `get_chunksize` / `iter_qmc_formatted` of
run_inference.py (compiled from its syntax tree).  Pins `tq_unrank`, `tq_numpy_choice_tail` (through
`random_combination_sample_via_index`), `get_chunksize`, `tq_format_qmc`, `tq_qmc_splits` -- and the oracle's
restatement oracle/qmc_format.py -- by reference outputs instead of hand-written expectations."""
from math import comb

import numpy as np
import pytest

from conftest import load_golden

from oracle import qmc_format as O
from tetrad_amd import combinations as C
from tetrad_amd import distributor as D
from tetrad_amd import qmc_format as P


@pytest.fixture(scope="module")
def g():
    return load_golden("host_f2f3")


def sample_checksum(q):
    q = np.asarray(q, np.uint64)
    packed = (q[:, 0] << np.uint64(24)) | (q[:, 1] << np.uint64(16)) | (q[:, 2] << np.uint64(8)) | q[:, 3]
    with np.errstate(over="ignore"):
        return (packed * (np.arange(len(q), dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64)


@pytest.mark.parametrize("T", [16, 64, 128, 256])
def test_unrank_equals_reference_index_to_combination(g, T):
    ranks, want = g[f"unrank_T{T}_ranks"], g[f"unrank_T{T}_quartets"]
    assert ranks[0] == 0 and ranks[-1] == comb(T, 4) - 1              # both ends of the rank space are in
    np.testing.assert_array_equal(C.unrank(ranks, T), want)
    # a consecutive range through the other entry of tq_unrank
    lo = int(ranks[len(ranks) // 2])
    n = min(50, comb(T, 4) - lo)
    np.testing.assert_array_equal(C.unrank(None, T, lo, n)[0], want[len(ranks) // 2])
    np.testing.assert_array_equal(C.unrank(None, T, lo, n), C.unrank(np.arange(lo, lo + n), T))


def test_sampler_equals_reference_in_both_choice_regimes(g):
    seen = set()
    for T, size, seed, tail in g["sample_cases"]:
        T, size, seed = int(T), int(size), int(seed)
        got = C.random_combination_sample_via_index(T, size, seed)
        assert got.shape == (size, 4) and got.dtype == np.uint32
        key = f"sample_T{T}_n{size}_s{seed}"
        if key in g:
            np.testing.assert_array_equal(got, g[key].astype(np.uint32))
        else:                                                         # stored as every 37th row + checksum
            np.testing.assert_array_equal(got[::37], g[key + "_every37"].astype(np.uint32))
            assert sample_checksum(got) == g[key + "_checksum"]
        seen.add(bool(tail))
        assert bool(tail) == (size > comb(T, 4) // 50)
    assert seen == {True, False}
    # the library's restatement of NumPy's tail shuffle must actually have been the code under test
    assert C._fast_choice_ok is True


def test_iter_chunks_random_chunking(g):
    chunks = list(C.iter_chunks_random(128, 2000, 300, 42))
    assert [len(c) for c in chunks] == g["chunks_T128_n2000_m300_s42_lens"].tolist()
    np.testing.assert_array_equal(np.concatenate(chunks), g["sample_T128_n2000_s42"].astype(np.uint32))


def test_get_chunksize_equals_reference(g):
    for i, nq in enumerate(g["chunksize_nquartets"]):
        for j, nc in enumerate(g["chunksize_ncores"]):
            assert D.get_chunksize(int(nq), int(nc)) == int(g["chunksize"][i, j]), (nq, nc)


@pytest.mark.parametrize("mode", ["sub", "full"])
def test_qmc_lines_equal_reference(g, mode, tmp_path):
    c1 = load_golden("c1_T16_S5000")
    q, rstat, rscor = c1["quartets"], c1[f"{mode}_rstat"], c1[f"{mode}_rscor"]
    tsv = bytes(g[f"qmc_{mode}_tsv"])
    # the TSV the reference's pandas call writes == the native TSV formatter on the same arrays
    assert D.format_tsv_bytes(q, rscor, rstat) == tsv
    f = tmp_path / "q.tsv"
    f.write_bytes(tsv)
    nonempty = 0
    for k, (w, ms, mr) in enumerate(g["qmc_settings"]):
        w, ms = int(w), int(ms)
        want = bytes(g[f"qmc_{mode}_lines_{k}"]).decode("ascii").split("\n")[:-1]
        want = [ln for ln in want if ln]
        nonempty += bool(want)
        # native formatter straight from the arrays, from the TSV, and the oracle's restatement
        assert [ln.decode() for ln in P.qmc_lines(q, rscor, rstat, w, ms, float(mr))] == want
        assert list(P.iter_qmc_formatted(f, w, ms, float(mr))) == want
        assert list(O.iter_qmc_formatted(f, w, ms, float(mr))) == want
        # tq_qmc_splits: the same rows as arrays (split a,b|c,d and the weight as its "%.5f" text reads back)
        from tetrad_amd import qmc
        splits, wts = qmc.qmc_splits(q, rscor, rstat, w, ms, float(mr))
        assert len(splits) == len(want)
        for s_, w_, ln in zip(splits[:200], wts[:200], want[:200]):
            left, rest = ln.split("|")
            right, wt = rest.split(":")
            assert [int(x) for x in left.split(",")] + [int(x) for x in right.split(",")] == s_.tolist()
            assert float(wt) == w_
    assert nonempty >= 10
