"""Quartet producers: host-side mirror of tetrad/src/combinations.py.

Same function names and arguments as the reference; the chunks are NumPy arrays u32[n,4] instead of lists / islices
of tuples (`infer_resolved_quartets` turns its `qrts` argument into such an array first thing anyway,
resolve_quartets.py:28, and `np.array(list(chunk))` accepts either).  The reference unranks every sampled index with
an O(nsamples) Python loop (`_index_to_combination`, :94-106: minutes for 5e6 quartets of 256 taxa); here the
library's native unranker does it (`tq_unrank`, host code; `tq_unrank_dev` is the device version the replicate loop
uses).  `random_combination_sample_via_index` makes the same single draw on the caller's Generator
(`rng.choice(total, size, replace=False)`, :113), so a project's random stream advances exactly as in the reference.
"""
from __future__ import annotations

import ctypes
from math import comb

import numpy as np
from numpy.random import default_rng


def unrank(ranks=None, nsamples: int = 0, first_rank: int = 0, count: int = 0) -> np.ndarray:
    """u32[n,4]: the 4-combinations of range(nsamples) with lexicographic ranks `ranks`, or with ranks
    first_rank .. first_rank+count-1 when `ranks` is None."""
    from . import _lib
    lib = _lib.load()
    if ranks is not None:
        r = np.ascontiguousarray(ranks, dtype=np.uint64).reshape(-1)
        n = r.shape[0]
        out = np.empty((n, 4), np.uint32)
        rc = lib.tq_unrank(r.ctypes.data, 0, n, int(nsamples), out.ctypes.data)
    else:
        n = int(count)
        out = np.empty((n, 4), np.uint32)
        rc = lib.tq_unrank(None, int(first_rank), n, int(nsamples), out.ctypes.data)
    if rc != 0:
        raise _lib.TetradHipError(rc, f"tq_unrank: rank out of range for C({nsamples},4)={comb(int(nsamples), 4)}")
    return out


_fast_choice_ok = None


def _fast_choice_raw(rng, pop: int, size: int, out=None):
    from . import _lib
    lib = _lib.load()
    bg = rng.bit_generator
    if out is None:
        out = np.empty(size, np.int64)
    with bg.lock:
        rc = lib.tq_numpy_choice_tail(bg.ctypes.bit_generator, int(pop), int(size), out.ctypes.data)
    if rc != 0:
        raise _lib.TetradHipError(rc, "tq_numpy_choice_tail")
    return out


def choice_without_replacement(rng, pop: int, size: int, out=None) -> np.ndarray:
    """`rng.choice(pop, size=size, replace=False)` -- the same int64 sample, the same Generator state afterwards --
    through the library's restatement of NumPy's tail shuffle (`tq_numpy_choice_tail`: sparse map + prefetch
    instead of an 8*pop-byte arange; about 3x faster for 1e6 of 1e7) when the call is in that regime
    (pop > 10 000, size > pop // 50, pop < 2^32 - 1) and the restatement has just proven itself against NumPy's
    own `Generator.choice` on this installation; NumPy's call otherwise."""
    global _fast_choice_ok
    pop, size = int(pop), int(size)
    if _fast_choice_ok is None:
        try:
            a = np.random.Generator(np.random.PCG64(20240229))
            b = np.random.Generator(np.random.PCG64(20240229))
            ok = True
            for p_, s_ in ((10_007, 201), (50_000, 50_000), (123_457, 40_000)):
                ok &= bool(np.array_equal(a.choice(p_, size=s_, replace=False), _fast_choice_raw(b, p_, s_)))
                ok &= a.bit_generator.state == b.bit_generator.state
            _fast_choice_ok = bool(ok)
        except Exception:
            _fast_choice_ok = False
    if _fast_choice_ok and 10_000 < pop < 0xFFFFFFFF and pop // 50 < size <= pop and isinstance(rng, np.random.Generator):
        return _fast_choice_raw(rng, pop, size, out)
    res = rng.choice(pop, size=size, replace=False)
    if out is not None:
        out[...] = res
        return out
    return res


def get_chunks_info(nsamples: int, max_chunk_size: int) -> list[tuple[int, int]]:
    """combinations.py:11-37: (start, end) rank ranges of the chunks of the full enumeration."""
    total = comb(nsamples, 4)
    return [(start, min(start + max_chunk_size, total)) for start in range(0, total, max_chunk_size)]


def get_combinations_from_chunk(nsamples: int, start: int, end: int) -> np.ndarray:
    """combinations.py:40-55: islice(combinations(range(nsamples), 4), start, end) as an array."""
    return unrank(None, nsamples, start, max(0, min(end, comb(nsamples, 4)) - start))


def iter_chunks_full(nsamples: int, max_size: int):
    """combinations.py:82-87."""
    for start, end in get_chunks_info(nsamples, max_size):
        yield get_combinations_from_chunk(nsamples, start, end)


def random_combination_sample_via_index(nsamples: int, size: int, rng) -> np.ndarray:
    """combinations.py:109-114: one `rng.choice(C(nsamples,4), size, replace=False)`, then unranking."""
    rng = default_rng(rng)
    return unrank(choice_without_replacement(rng, comb(nsamples, 4), size), nsamples)


def iter_chunks_random(nsamples: int, size: int, max_size: int, rng):
    """combinations.py:117-121 (a generator, like the reference's: the draw happens at the first next())."""
    qrts = random_combination_sample_via_index(nsamples, size, rng)
    for i in range(0, len(qrts), max_size):
        yield qrts[i: i + max_size]
