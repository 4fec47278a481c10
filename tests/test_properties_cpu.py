"""Property tests (hypothesis) of the CPU-side pieces: oracle C kernels vs the Python loops,
unranking vs itertools, record packing, shard bounds, span table."""
from itertools import combinations, islice
from math import comb

import numpy as np
from hypothesis import given, settings, strategies as st

from tetrad_amd import bootstrap, distributor as D, synth


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 300), st.integers(0, 2**31 - 1), st.floats(0.0, 0.9), st.integers(1, 12))
def test_count_kernels_match_python_loops(S, seed, miss, mean_run):
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    seqs = rng.integers(0, 4, size=(4, S), dtype=np.uint8)
    seqs[rng.random(seqs.shape) < miss] = 78
    loc = np.repeat(np.arange(S, dtype=np.uint32), 1 + rng.poisson(mean_run, size=S))[:S]
    mask = np.sum(seqs >= 78, axis=0) + (np.sum(seqs == seqs[0], axis=0) == 4)
    for sub in (False, True):
        fn = orc.subsample_chunk_to_matrices if sub else orc.full_chunk_to_matrices
        m = fn(seqs, loc, mask)
        np.testing.assert_array_equal(m, orc.chunk_to_matrices_py(seqs, loc, mask, sub))
        assert m[0].sum() == m[1].sum() == m[2].sum()
        if sub:
            assert m[0].sum() <= len(np.unique(loc))


@settings(max_examples=30, deadline=None)
@given(st.integers(4, 40), st.integers(0, 2**31 - 1))
def test_unranking_matches_itertools(T, seed):
    total = comb(T, 4)
    rng = np.random.default_rng(seed)
    ranks = rng.integers(0, total, size=20)
    got = synth.unrank_quartets(ranks, T)
    for r, q in zip(ranks, got):
        assert tuple(int(x) for x in q) == next(islice(combinations(range(T), 4), int(r), None))


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 10**7), st.integers(1, 64))
def test_shard_bounds(Q, world):
    b = D.shard_bounds(Q, world)
    assert b[0][0] == 0 and b[-1][1] == Q and len(b) == world
    sizes = [hi - lo for lo, hi in b]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1 and sum(sizes) == Q


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 400), st.integers(0, 2**31 - 1))
def test_spans_partition_the_sites(S, seed):
    from oracle import resample as R
    rng = np.random.default_rng(seed)
    loc = np.repeat(np.arange(S), 1 + rng.poisson(3, size=S))[:S]
    m = np.stack([loc, np.arange(S)], axis=1)
    sp = bootstrap.get_spans(m)
    assert sp[0, 0] == 0 and sp[-1, 1] == S and (sp[1:, 0] == sp[:-1, 1]).all() and (sp[:, 1] > sp[:, 0]).all()
    if len(np.unique(loc)) > 1:
        np.testing.assert_array_equal(sp, R.get_spans(m))


def test_hdf5_branch_of_load_database_with_a_stand_in_h5py(tmp_path, monkeypatch):
    """`infer_resolved_quartets` opens the reference's database as `h5py.File(path, 'r', swmr=True)` and reads the
    datasets `tmparr` / `tmpmap` whole (resolve_quartets.py:33-35).  h5py is not installed in this image, so the branch
    is exercised with a stand-in module that accepts exactly that call shape and serves the arrays of an .npz kept
    beside the .hdf5 path (the loader's own logic -- suffix dispatch, caching on (path, mtime, size), the replicate token
    -- is what is under test; no compute call, so no GPU)."""
    import sys
    import types
    calls = []

    class File:
        def __init__(self, path, mode, swmr=False):
            calls.append((str(path), mode, swmr))
            self._z = np.load(str(path) + ".npz", allow_pickle=False)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self._z.close()

        def __getitem__(self, name):
            return self._z[name]

    fake = types.ModuleType("h5py")
    fake.File = File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    from tetrad_amd import resolve_quartets as RQ
    rng = np.random.default_rng(3)
    tmparr = rng.integers(0, 4, size=(6, 500), dtype=np.uint8)
    tmpmap = np.stack([np.arange(500) // 5, np.arange(500)], axis=1).astype(np.uint32)
    db = tmp_path / "proj.snps.hdf5"
    db.write_bytes(b"\x89HDF\r\n\x1a\n")                     # the path must exist: the cache key is (path, mtime, size)
    np.savez(str(db) + ".npz", tmparr=tmparr, tmpmap=tmpmap)
    monkeypatch.setattr(RQ, "_db_cache", {})
    a, m, token = RQ.load_database(db, with_token=True)
    np.testing.assert_array_equal(a, tmparr)
    np.testing.assert_array_equal(m, tmpmap)
    assert calls == [(str(db), "r", True)]
    a2, m2, token2 = RQ.load_database(db, with_token=True)    # second chunk of the same replicate: cached, same token
    assert a2 is a and token2 == token and len(calls) == 1
    db.write_bytes(b"\x89HDF\r\n\x1a\n" + b"x")               # the file changed (next replicate): read again, new token
    a3, _, token3 = RQ.load_database(db, with_token=True)
    assert token3 != token and len(calls) == 2
