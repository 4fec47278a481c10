"""The native text producers of the C ABI (tq_format_tsv / tq_format_qmc, host code) against the
Python formulas of the reference: run_inference.py:233-234 (`to_csv(float_format='%.6f')`, i.e.
`'%.6f' % x`) and :254-305 (`iter_qmc_formatted`, restated in oracle/qmc_format.py)."""
import numpy as np
import pytest

from tetrad_amd import distributor as D
from oracle import qmc_format as OQ
from tetrad_amd import qmc_format as Q


def _py_tsv(q, sc, st):
    return "".join("%d\t%d\t%d\t%d\t%.6f\t%.6f\t%.6f\t%d\t%d\n" % (*a, *b, *c)
                   for a, b, c in zip(q.tolist(), sc.tolist(), st.tolist()))


def _rows(n, seed):
    rng = np.random.default_rng(seed)
    q = np.sort(rng.integers(0, 300, size=(n, 4)), axis=1).astype(np.uint32)
    sc = rng.gamma(2.0, 20.0, size=(n, 3))
    st = np.stack([rng.integers(0, 3, size=n), rng.integers(0, 60000, size=n)], axis=1).astype(np.uint32)
    return q, sc, st


def test_tsv_matches_percent_formatting():
    q, sc, st = _rows(20000, 1)
    assert D.format_tsv(q, sc, st) == _py_tsv(q, sc, st)


def test_tsv_awkward_values():
    """ties at the 7th decimal, values that round up into a new digit, tiny, huge and zero scores"""
    vals = np.array([0.0, 0.001, 0.0000005, 0.0000015, 0.0000025, 0.9999995, 0.99999949999, 9.9999995, 123456.7890125,
                     2.5e-7, 1e-300, 4.1e9, 3.999999e9, 1.5e15, 1e22, 1.7976931348623157e308, 0.1234565, 0.1234575,
                     1.0000005, 1.0000015, 2 ** -20, 2 ** -21 + 2 ** -40, 1 / 3, 2 / 3, 1e6 + 0.0000005])
    vals = np.concatenate([vals, np.arange(1, 2001) * 5e-7])          # every multiple of 5e-7 up to 1e-3
    n = len(vals)
    q = np.tile(np.array([[0, 1, 2, 4294967295]], np.uint32), (n, 1))
    with np.errstate(over="ignore"):
        sc = np.stack([vals, vals[::-1], vals * 3.0], axis=1)      # the last column reaches inf
    st = np.tile(np.array([[2, 4294967295]], np.uint32), (n, 1))
    assert D.format_tsv(q, sc, st) == _py_tsv(q, sc, st)


def test_tsv_empty_and_bytes():
    q, sc, st = _rows(0, 2)
    assert D.format_tsv(q, sc, st) == ""
    q, sc, st = _rows(5, 3)
    assert D.format_tsv_bytes(q, sc, st) == _py_tsv(q, sc, st).encode()


@pytest.mark.parametrize("weights", [0, 1, 2, 3])
@pytest.mark.parametrize("min_snps,min_ratio", [(0, 1.0), (500, 1.0), (0, 1.6), (20000, 1.2)])
def test_qmc_lines_match_file_based_mirror(tmp_path, weights, min_snps, min_ratio):
    q, sc, st = _rows(5000, 10 + weights)
    sc[::97] = 0.001                       # zero-data style rows
    st[::97, 1] = 0
    sc[5::211, 0] = 0.0                    # a zero smallest score: ratio is defined as 1
    f = tmp_path / "q.tsv"
    f.write_text(_py_tsv(q, sc, st))
    want = list(OQ.iter_qmc_formatted(f, weights, min_snps, min_ratio))
    assert list(Q.iter_qmc_formatted(f, weights, min_snps, min_ratio)) == want
    got = [b.decode() for b in Q.qmc_lines(q, sc, st, weights, min_snps, min_ratio)]
    assert got == want


def test_qmc_rejects_unknown_strategy():
    q, sc, st = _rows(3, 4)
    with pytest.raises(ValueError):
        Q.qmc_lines(q, sc, st, weights=7)


def test_write_qmc_from_arrays_is_a_seeded_permutation(tmp_path):
    q, sc, st = _rows(400, 5)
    n = Q.write_qmc_from_arrays(q, sc, st, tmp_path / "a.txt", weights=2, seed=9)
    Q.write_qmc_from_arrays(q, sc, st, tmp_path / "b.txt", weights=2, seed=9)
    a = (tmp_path / "a.txt").read_text()
    assert a == (tmp_path / "b.txt").read_text()
    lines = [b.decode() for b in Q.qmc_lines(q, sc, st, 2)]
    assert n == len(lines) and sorted(a.splitlines()) == sorted(lines) and a.splitlines() != lines


def test_tsv_pieces_concatenate_to_the_same_text():
    q, sc, st = _rows(30_000, 8)
    want = D.format_tsv_bytes(q, sc, st)
    for threads, rows in ((4, 7_000), (8, 1 << 18), (1, 100)):
        assert b"".join(D.format_tsv_pieces(q, sc, st, threads=threads, rows_per_piece=rows)) == want
