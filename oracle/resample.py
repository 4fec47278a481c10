"""oracle/resample.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the bootstrap-replicate producer of the hot path:
tetrad/jit/get_spans.py:11-48, tetrad/jit/resample.py:7-64, tetrad/jit/resolve_ambigs.py:12-36
(duplicated in tetrad/src/jitted.py:21-145) and the driver tetrad/src/run_inference.py:99-143.

Pinning: the span table is deterministic and pinned bit-for-bit by tests/golden/resample_*.npz.
The two randomised functions draw from numba's own Mersenne twister in the real reference; that
stream cannot be reproduced without numba ("RNG-stream parity unpinned").  The
golden files hold the reference's code executed with NumPy's legacy global RNG in numba's place
(tests/golden/make_golden.py); this restatement makes the same NumPy calls in the same order, so it
matches them bit-for-bit -- which pins the LOGIC (block layout, locus ordinals, which codes may be
resolved to what), not the stream.  The device bootstrap (tq_bootstrap) uses its own counter-based
stream and is tested against the structural properties this module defines.
"""
from __future__ import annotations

import numpy as np

# tetrad/src/utils.py:14-21 -- ambiguity code, resolution 1, resolution 2
GETCONS = np.array([
    [82, 71, 65],   # R -> G / A
    [75, 71, 84],   # K -> G / T
    [83, 71, 67],   # S -> G / C
    [89, 84, 67],   # Y -> T / C
    [87, 84, 65],   # W -> T / A
    [77, 67, 65],   # M -> C / A
], dtype=np.uint8)


def get_spans(maparr: np.ndarray) -> np.ndarray:
    """jit/get_spans.py:11-48: [start,end) site ranges of the loci of a (sorted) snpsmap."""
    maparr = np.asarray(maparr)
    sidx = maparr[0, 0]
    locs = np.unique(maparr[:, 0])
    nlocs = locs.size
    spans = np.zeros((nlocs, 2), np.int64)
    lidx = 0
    for idx in range(maparr.shape[0]):                    # :32
        eidx = maparr[idx, 0]
        if eidx != sidx:                                  # :38
            if not lidx:
                spans[lidx] = (0, idx)                    # :42
            else:
                spans[lidx] = (spans[lidx - 1, 1], idx)   # :46
            lidx += 1
            sidx = locs[lidx]
    spans[-1] = (spans[-2, 1], maparr[-1, -1] + 1)        # :52
    return spans


def get_nsites(spans: np.ndarray, lidxs: np.ndarray) -> int:
    """jit/resample.py:7-17."""
    width = 0
    for idx in range(lidxs.size):
        width += spans[lidxs[idx], 1] - spans[lidxs[idx], 0]
    return int(width)


def resample(seqarr: np.ndarray, spans: np.ndarray, lidxs: np.ndarray, seed: int):
    """jit/resample.py:20-64 (NumPy legacy RNG in numba's place)."""
    np.random.seed(seed)                                                  # :28
    arrlen = get_nsites(spans, lidxs)
    tmparr = np.zeros((seqarr.shape[0], arrlen), dtype=np.uint8)
    tmpmap = np.zeros((arrlen, 2), dtype=np.uint32)
    tmpmap[:, 1] = np.arange(arrlen, dtype=np.uint32)
    cidx = 0
    for idx, lidx in enumerate(lidxs):                                    # :40
        start, end = spans[lidx]
        cols = seqarr[:, start:end]
        col_idxs = np.random.choice(cols.shape[1], cols.shape[1], replace=False)   # :49
        tmparr[:, cidx:cidx + cols.shape[1]] = cols[:, col_idxs]          # :54
        tmpmap[cidx: cidx + cols.shape[1], 0] = idx                       # :58
        cidx += cols.shape[1]
    return tmparr, tmpmap


def resolve_ambigs(tmpseq: np.ndarray, seed: int) -> np.ndarray:
    """jit/resolve_ambigs.py:12-36 (in place, like the reference)."""
    np.random.seed(seed)                                                  # :22
    for aidx in range(6):
        ambig, res1, res2 = GETCONS[aidx]
        idx, idy = np.where(tmpseq == ambig)                              # :29
        halfmask = np.random.binomial(n=1, p=0.5, size=idx.shape[0])      # :30
        for col in range(idx.shape[0]):
            tmpseq[idx[col], idy[col]] = res1 if halfmask[col] else res2  # :31-35
    return tmpseq


def recode(tmparr: np.ndarray) -> np.ndarray:
    """run_inference.py:133-136 / write_database.py:164-167: A,C,G,T -> 0,1,2,3 (in place)."""
    tmparr[tmparr == 65] = 0
    tmparr[tmparr == 67] = 1
    tmparr[tmparr == 71] = 2
    tmparr[tmparr == 84] = 3
    return tmparr


def resample_tmp_database(seqarr: np.ndarray, spans: np.ndarray, rng):
    """run_inference.py:99-143 without the HDF5 I/O: returns (tmparr, tmpmap, lidxs, seed1, seed2)."""
    rng = np.random.default_rng(rng)                                      # :105
    nloci = spans.shape[0]
    lidxs = rng.choice(nloci, nloci, replace=True)                        # :117
    seed1 = int(rng.integers(2**31))
    tmparr, tmpmap = resample(seqarr, spans, lidxs, seed=seed1)           # :120
    seed2 = int(rng.integers(2**31))
    tmparr = resolve_ambigs(tmparr, seed=seed2)                           # :123
    return recode(tmparr), tmpmap, lidxs, seed1, seed2


# ---- structural properties every valid replicate has (used to check the device bootstrap) ------
def check_replicate(seqarr, spans, lidxs, tmparr, tmpmap):
    """Raise AssertionError unless (tmparr, tmpmap) is a valid outcome of resample + resolve + recode
    for the resampled loci `lidxs`, whatever the random stream was."""
    T = seqarr.shape[0]
    widths = (spans[lidxs, 1] - spans[lidxs, 0]).astype(np.int64)
    S = int(widths.sum())
    assert tmparr.shape == (T, S) and tmpmap.shape == (S, 2)
    np.testing.assert_array_equal(tmpmap[:, 1], np.arange(S))
    np.testing.assert_array_equal(tmpmap[:, 0], np.repeat(np.arange(len(lidxs)), widths))
    allowed = {int(a): {int(r1), int(r2)} for a, r1, r2 in GETCONS}
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    off = 0
    for lidx, w in zip(lidxs, widths):
        src = seqarr[:, spans[lidx, 0]:spans[lidx, 1]]
        out = tmparr[:, off:off + w]
        off += w
        # every output column must be some source column of the same locus (after resolution and
        # recoding), and the assignment must be a bijection: greedy matching on exact candidates
        used = np.zeros(w, bool)
        for j in range(w):
            ok = None
            for k in range(w):
                if used[k]:
                    continue
                good = True
                for t in range(T):
                    s, o = int(src[t, k]), int(out[t, j])
                    if s in code:
                        good = o == code[s]
                    elif s in allowed:
                        good = o in {code[x] for x in allowed[s]}
                    else:
                        good = o > 3 or o == s       # missing stays missing (any byte > 3)
                    if not good:
                        break
                if good:
                    ok = k
                    break
            assert ok is not None, "output column is not a (resolved) column of its source locus"
            used[ok] = True
