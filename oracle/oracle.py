"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of tetrad's per-quartet hot path
(reference: tetrad/src/resolve_quartets.py:191-265 and the kernels :42-104).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``tetrad_amd/`` does: the product
path is the HIP library and fails loudly when it is missing.

Structure mirrors the reference worker line by line: an interpreted per-quartet
loop (the reference's driver is *not* jitted), NumPy mask reductions, a compiled
serial count loop (C here, numba in the reference) and ``numpy.linalg.svd`` /
``numpy.linalg.matrix_rank`` -- the third-party arithmetic the reference itself
calls (numpy is unpinned by the reference: setup.py:45; fixtures were made with
NumPy 2.2.6 / OpenBLAS 0.3.29).

Pinning: ``tests/test_oracle_golden.py`` checks this module against
``tests/golden/*.npz`` which hold outputs of the reference's own functions
(``tests/golden/make_golden.py``).

Documented deviations (inputs the reference leaves undefined):
  * zero-data quartets: the reference draws ``np.random.randint(3)`` unseeded
    (:231); the oracle returns topology 0 and sets flag bit 0.
  * genotype bytes 4..77 (outside the reference's precondition, would index
    outside the 16x16 matrix) are treated as masked.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libtetrad_oracle.so"
_lib = None

FLAG_ZERO_DATA = 1
FLAG_DEGENERATE = 2
#: two lowest scores closer than this (relative to the largest singular value
#: of the three flattenings) => topology is decided by rounding noise of the
#: SVD implementation, not by the data ("hard parts").
DEGENERATE_REL_GAP = 1e-9


def build(force: bool = False) -> Path:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = _HERE / "count_matrices.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-o", str(_LIB_PATH), str(src)]
        )
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        lib = ctypes.CDLL(str(_LIB_PATH))
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        lib.oracle_quartet_mask.argtypes = [u8p, ctypes.c_int64, u8p]
        lib.oracle_quartet_mask.restype = None
        for name in ("oracle_full_chunk_to_matrices", "oracle_subsample_chunk_to_matrices"):
            fn = getattr(lib, name)
            fn.argtypes = [u8p, ctypes.c_int64, u32p, u8p, u32p]
            fn.restype = None
        lib.oracle_quartet_to_matrices.argtypes = [
            u8p, ctypes.c_int64, u32p, u32p, ctypes.c_int, u8p, u32p]
        lib.oracle_quartet_to_matrices.restype = None
        _lib = lib
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


# --------------------------------------------------------------------------
# kernel-level functions (resolve_quartets.py:42-104)
# --------------------------------------------------------------------------
def full_chunk_to_matrices(seqs: np.ndarray, locus: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """resolve_quartets.py:76-104.  seqs u8[4,S], locus u32[S], mask[S] -> u32[3,16,16]."""
    return _chunk_to_matrices(seqs, locus, mask, subsample=False)


def subsample_chunk_to_matrices(seqs: np.ndarray, locus: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """resolve_quartets.py:42-73.  Counts a site only if unmasked and its locus differs
    from the locus of the last *counted* site (last_loc starts at uint32(-1))."""
    return _chunk_to_matrices(seqs, locus, mask, subsample=True)


def _chunk_to_matrices(seqs, locus, mask, subsample):
    lib = _load()
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    assert seqs.ndim == 2 and seqs.shape[0] == 4
    S = seqs.shape[1]
    locus = np.ascontiguousarray(locus, dtype=np.uint32)
    m8 = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    # bytes outside 0..3 in an unmasked site are undefined in the reference; mask them
    m8 = m8 | (seqs > 3).any(axis=0).astype(np.uint8)
    mats = np.zeros((3, 16, 16), dtype=np.uint32)
    fn = lib.oracle_subsample_chunk_to_matrices if subsample else lib.oracle_full_chunk_to_matrices
    fn(_p(seqs, ctypes.c_uint8), S, _p(locus, ctypes.c_uint32), _p(m8, ctypes.c_uint8),
       _p(mats, ctypes.c_uint32))
    return mats


def chunk_to_matrices_py(seqs, locus, mask, subsample):
    """Pure-Python/NumPy version of the two count kernels (small cases only);
    used to cross-check the C restatement."""
    mats = np.zeros((3, 16, 16), dtype=np.uint32)
    last_loc = 0xFFFFFFFF                                 # :58
    for idx in range(seqs.shape[1]):                      # :59 / :92
        if not mask[idx]:
            if (not subsample) or int(locus[idx]) != last_loc:
                i = seqs[:, idx].astype(np.int64)
                mats[0, 4 * i[0] + i[1], 4 * i[2] + i[3]] += 1
                last_loc = int(locus[idx])
    x = 0
    for y in (0, 4, 8, 12):                               # :68-72
        for z in (0, 4, 8, 12):
            mats[1, y:y + 4, z:z + 4] = mats[0, x].reshape(4, 4)
            mats[2, y:y + 4, z:z + 4] = mats[0, x].reshape(4, 4).T
            x += 1
    return mats


# --------------------------------------------------------------------------
# worker-level function (resolve_quartets.py:191-265)
# --------------------------------------------------------------------------
def score_from_cmats(cmats: np.ndarray):
    """resolve_quartets.py:237-251 for one quartet with nsnps > 0.

    Returns (svds f64[3,16], rank f64[3], scores f64[3], topo int)."""
    svds = np.zeros((3, 16), dtype=np.float64)
    rank = np.zeros(3, dtype=np.float64)
    for test in range(3):                                 # :241-243
        m = cmats[test].astype(np.float64)
        svds[test] = np.linalg.svd(m)[1]
        rank[test] = np.linalg.matrix_rank(m)
    minrank = int(min(10, rank.min()))                    # :246
    scor = np.zeros(3, dtype=np.float64)
    for test in range(3):                                 # :247-248
        scor[test] = np.sqrt(np.sum(svds[test, minrank:] ** 2))
    return svds, rank, scor, int(np.argmin(scor))         # :251


def degenerate_flag(scores: np.ndarray, svds: np.ndarray) -> bool:
    """True when the argmin is decided by SVD rounding noise (two lowest scores
    within DEGENERATE_REL_GAP * sigma_max of each other, exact ties included)."""
    s = np.sort(scores)
    return bool((s[1] - s[0]) <= DEGENERATE_REL_GAP * svds.max())


def new_infer_resolved_quartets(tmparr, tmpmap, quartets, subsample_snps, debug: bool = False):
    """resolve_quartets.py:191-265.

    tmparr u8[T,S], tmpmap u32[S,2], quartets u32[Q,4] ->
    (quartets, rstat u32[Q,2], rscor f64[Q,3]); with ``debug`` also a dict of
    cmats u32[Q,3,16,16], svds f64[Q,3,16], rank i32[Q,3], flags u8[Q]."""
    lib = _load()
    tmparr = np.ascontiguousarray(tmparr, dtype=np.uint8)
    quartets = np.ascontiguousarray(quartets, dtype=np.uint32).reshape(-1, 4)
    locus = np.ascontiguousarray(np.asarray(tmpmap)[:, 0], dtype=np.uint32)
    Q = quartets.shape[0]
    S = tmparr.shape[1]
    rscor = np.zeros((Q, 3), dtype=np.float64)            # :205
    rstat = np.zeros((Q, 2), dtype=np.uint32)             # :206
    flags = np.zeros(Q, dtype=np.uint8)
    if debug:
        d_cmats = np.zeros((Q, 3, 16, 16), dtype=np.uint32)
        d_svds = np.zeros((Q, 3, 16), dtype=np.float64)
        d_rank = np.zeros((Q, 3), dtype=np.int32)
    scratch = np.empty(5 * max(S, 1), dtype=np.uint8)
    cmats = np.zeros((3, 16, 16), dtype=np.uint32)
    for qidx in range(Q):                                 # :208
        # :212-223 in one compiled call (row gather, two masks, count loop)
        lib.oracle_quartet_to_matrices(
            _p(tmparr, ctypes.c_uint8), S, _p(locus, ctypes.c_uint32),
            _p(quartets[qidx], ctypes.c_uint32), int(bool(subsample_snps)),
            _p(scratch, ctypes.c_uint8), _p(cmats, ctypes.c_uint32))
        nsnps = cmats[0].sum()                            # :226
        if not nsnps:                                     # :230-232
            rstat[qidx, 0] = 0                            # reference: unseeded randint(3)
            rscor[qidx] = 0.001
            flags[qidx] |= FLAG_ZERO_DATA
        else:
            svds, rank, scor, topo = score_from_cmats(cmats)
            rscor[qidx] = scor
            rstat[qidx, 0] = topo
            if degenerate_flag(scor, svds):
                flags[qidx] |= FLAG_DEGENERATE
            if debug:
                d_svds[qidx] = svds
                d_rank[qidx] = rank
        rstat[qidx, 1] = nsnps                            # :264
        if debug:
            d_cmats[qidx] = cmats
    if debug:
        return quartets, rstat, rscor, dict(cmats=d_cmats, svds=d_svds, rank=d_rank, flags=flags)
    return quartets, rstat, rscor                         # :265


def new_infer_resolved_quartets_batched(tmparr, tmpmap, quartets, subsample_snps, chunk: int = 4096):
    """What a tuned CPU implementation of the same path would do (the second CPU figure SURVEY.md
    8d asks for, so that the GPU/CPU ratio is not inflated by interpreter overhead): compiled count
    loop per quartet, then ONE values-only LAPACK SVD per matrix for a whole chunk
    (`np.linalg.svd(..., compute_uv=False)` on a [n,3,16,16] stack) instead of the reference's full
    SVD + second SVD inside matrix_rank (:242-243), rank and scores vectorised.  Same rules
    (:243-251); values agree with `new_infer_resolved_quartets` to rounding."""
    lib = _load()
    tmparr = np.ascontiguousarray(tmparr, dtype=np.uint8)
    quartets = np.ascontiguousarray(quartets, dtype=np.uint32).reshape(-1, 4)
    locus = np.ascontiguousarray(np.asarray(tmpmap)[:, 0], dtype=np.uint32)
    Q, S = quartets.shape[0], tmparr.shape[1]
    rscor = np.zeros((Q, 3), dtype=np.float64)
    rstat = np.zeros((Q, 2), dtype=np.uint32)
    scratch = np.empty(5 * max(S, 1), dtype=np.uint8)
    eps = np.finfo(np.float64).eps
    for q0 in range(0, Q, chunk):
        n = min(chunk, Q - q0)
        cm = np.zeros((n, 3, 16, 16), dtype=np.uint32)
        for i in range(n):
            lib.oracle_quartet_to_matrices(
                _p(tmparr, ctypes.c_uint8), S, _p(locus, ctypes.c_uint32),
                _p(quartets[q0 + i], ctypes.c_uint32), int(bool(subsample_snps)),
                _p(scratch, ctypes.c_uint8), _p(cm[i], ctypes.c_uint32))
        nsnps = cm[:, 0].reshape(n, -1).sum(axis=1)
        sv = np.linalg.svd(cm.astype(np.float64), compute_uv=False)            # [n,3,16], descending
        rank = (sv > sv[:, :, :1] * (16 * eps)).sum(axis=2)                     # matrix_rank's rule
        minrank = np.minimum(10, rank.min(axis=1))
        keep = np.arange(16)[None, None, :] >= minrank[:, None, None]
        scor = np.sqrt(np.where(keep, sv * sv, 0.0).sum(axis=2))
        zero = nsnps == 0
        scor[zero] = 0.001
        rscor[q0:q0 + n] = scor
        rstat[q0:q0 + n, 0] = np.where(zero, 0, np.argmin(scor, axis=1))
        rstat[q0:q0 + n, 1] = nsnps
    return quartets, rstat, rscor


def new_infer_resolved_quartets_numpy(tmparr, tmpmap, quartets, subsample_snps):
    """Same as above but with the reference's NumPy temporaries spelled out
    (:212-223: fancy-index row gather, two mask reductions, then the count
    kernel).  This is the 'reference-faithful interpreted' CPU baseline."""
    tmparr = np.asarray(tmparr)
    quartets = np.asarray(quartets, dtype=np.uint32).reshape(-1, 4)
    locus = np.ascontiguousarray(np.asarray(tmpmap)[:, 0], dtype=np.uint32)
    Q = quartets.shape[0]
    rscor = np.zeros((Q, 3), dtype=np.float64)
    rstat = np.zeros((Q, 2), dtype=np.uint32)
    for qidx in range(Q):
        sidx = quartets[qidx]
        seqs = tmparr[sidx, :]                            # :212
        nmask0 = np.sum(seqs >= 78, axis=0)               # :216
        nmask1 = np.sum(seqs == seqs[0], axis=0) == 4     # :218
        if subsample_snps:                                # :220-223
            cmats = subsample_chunk_to_matrices(seqs, locus, nmask0 + nmask1)
        else:
            cmats = full_chunk_to_matrices(seqs, locus, nmask0 + nmask1)
        nsnps = cmats[0].sum()
        if not nsnps:
            rstat[qidx, 0] = 0
            rscor[qidx] = 0.001
        else:
            _, _, scor, topo = score_from_cmats(cmats)
            rscor[qidx] = scor
            rstat[qidx, 0] = topo
        rstat[qidx, 1] = nsnps
    return quartets, rstat, rscor


if __name__ == "__main__":
    print(build(force=True))
