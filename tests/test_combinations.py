"""Host-side mirror of the reference's quartet producers (tetrad/src/combinations.py) and the native unranker
`tq_unrank` (host code of the C ABI: runs without a GPU)."""
from itertools import combinations, islice
from math import comb

import numpy as np
import pytest

from tetrad_amd import combinations as C
from tetrad_amd import synth


def _rank_of(q, n):
    """lexicographic rank of a sorted 4-tuple among the 4-combinations of range(n) (combinatorial number system):
    the inverse map, computed independently of any unranker"""
    a, b, c, d = (int(x) for x in q)
    r = sum(comb(n - x - 1, 3) for x in range(a))
    r += sum(comb(n - x - 1, 2) for x in range(a + 1, b))
    r += sum(comb(n - x - 1, 1) for x in range(b + 1, c))
    return r + (d - c - 1)


@pytest.mark.parametrize("n", [4, 5, 9, 16])
def test_unrank_is_itertools_order(n):
    want = np.array(list(combinations(range(n), 4)), np.uint32)
    np.testing.assert_array_equal(C.unrank(np.arange(comb(n, 4)), n), want)
    np.testing.assert_array_equal(C.unrank(None, n, 0, comb(n, 4)), want)
    np.testing.assert_array_equal(np.concatenate(list(C.iter_chunks_full(n, 7))), want)
    assert C.get_chunks_info(n, 7)[-1][1] == comb(n, 4)


def test_unrank_random_ranks_large_t():
    rng = np.random.default_rng(0)
    for n in (64, 256, 1700):
        r = rng.choice(comb(n, 4), size=300, replace=False)
        got = C.unrank(r, n)
        assert (np.diff(got.astype(np.int64), axis=1) > 0).all() and got.max() < n
        assert [_rank_of(q, n) for q in got] == [int(i) for i in r]
        np.testing.assert_array_equal(got, synth.unrank_quartets(r, n))
        # first and last combination
    np.testing.assert_array_equal(C.unrank([0, comb(256, 4) - 1], 256), [[0, 1, 2, 3], [252, 253, 254, 255]])
    start = comb(100, 4) - 5
    np.testing.assert_array_equal(C.get_combinations_from_chunk(100, start, start + 50),
                                  np.array(list(islice(combinations(range(100), 4), start, None)), np.uint32))


def test_random_sample_makes_the_reference_draw():
    """One rng.choice on the caller's Generator (combinations.py:113): same sample, same Generator state after."""
    a, b = np.random.default_rng(5), np.random.default_rng(5)
    q = C.random_combination_sample_via_index(40, 500, a)
    idx = b.choice(comb(40, 4), size=500, replace=False)
    allq = np.array(list(combinations(range(40), 4)), np.uint32)
    np.testing.assert_array_equal(q, allq[idx])
    assert a.bit_generator.state == b.bit_generator.state
    chunks = list(C.iter_chunks_random(40, 500, 128, np.random.default_rng(5)))
    assert [len(c) for c in chunks] == [128, 128, 128, 116]
    np.testing.assert_array_equal(np.concatenate(chunks), q)


def test_unrank_rejects_ranks_beyond_the_space():
    from tetrad_amd._lib import TetradHipError
    with pytest.raises(TetradHipError):
        C.unrank([comb(10, 4)], 10)
    with pytest.raises(TetradHipError):
        C.unrank(None, 10, comb(10, 4) - 1, 2)


@pytest.mark.parametrize("pop,size", [(10_001, 201), (10_001, 10_001), (20_001, 20_000), (65_537, 1_400), (123_457, 40_000),
                                      (635_376, 100_000), (2_000_003, 40_001), (10_668_000, 1_000_000)])
def test_fast_choice_is_numpys_choice_bit_for_bit(pop, size):
    """tq_numpy_choice_tail against Generator.choice itself (NumPy's tail-shuffle regime: pop > 10 000 and
    size > pop // 50): the same int64 sample AND the same Generator state afterwards, so that a project's random
    stream is unchanged; two calls in a row stay in step."""
    assert pop > 10_000 and size > pop // 50
    for seed in (0, 12345):
        a, b = np.random.default_rng(seed), np.random.default_rng(seed)
        for _ in range(2):
            want = a.choice(pop, size=size, replace=False)
            got = C.choice_without_replacement(b, pop, size)
            assert got.dtype == np.int64
            np.testing.assert_array_equal(got, want)
            assert a.bit_generator.state == b.bit_generator.state
    assert C._fast_choice_ok is True


def test_fast_choice_falls_back_outside_its_regime():
    """small populations / small samples (NumPy uses Floyd's algorithm there) and other bit generators go through
    NumPy itself: still the same sample and state"""
    for pop, size in ((5000, 4000), (1_000_000, 100), (10_668_000, 213_360)):
        a, b = np.random.default_rng(3), np.random.default_rng(3)
        np.testing.assert_array_equal(C.choice_without_replacement(b, pop, size), a.choice(pop, size=size, replace=False))
        assert a.bit_generator.state == b.bit_generator.state
    a = np.random.Generator(np.random.Philox(9))
    b = np.random.Generator(np.random.Philox(9))
    np.testing.assert_array_equal(C.choice_without_replacement(b, 50_000, 20_000), a.choice(50_000, size=20_000, replace=False))
    assert a.bit_generator.state["state"]["counter"].tolist() == b.bit_generator.state["state"]["counter"].tolist()
