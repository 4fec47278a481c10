"""NumPy model of the device-side singular-value routine (tetrad_amd/csrc).

Test helper only: it mirrors, step for step, what one 16-lane group of a
wavefront does in the HIP kernel (one matrix column per lane, XOR-partner
parallel ordering, one-sided Hestenes-Jacobi rotations in f64), vectorised over
a batch of matrices.  It exists so that the schedule, tolerance and rank rule
can be validated against numpy.linalg on the CPU before any GPU time is spent;
the product never imports it.
"""
from __future__ import annotations

import numpy as np

EPS = np.finfo(np.float64).eps
TOL = 2.0 ** -50          # rotate while g^2 > TOL^2 * a*b
EARLY = 1e-5              # a sweep whose largest pre-rotation |g|/sqrt(a*b) is below this is the last
MAX_SWEEPS = 30


def jacobi_singular_values(mats: np.ndarray, tol: float = TOL, max_sweeps: int = MAX_SWEEPS,
                           return_sweeps: bool = False):
    """mats: [N,16,16] (any real dtype).  Returns singular values [N,16], descending."""
    A = np.array(mats, dtype=np.float64)          # A[n, row, col]; lane j owns column j
    N = A.shape[0]
    lanes = np.arange(16)
    sweeps = np.zeros(N, dtype=np.int32)
    active = np.ones(N, dtype=bool)
    for sweep in range(max_sweeps):
        nrm = np.einsum("nrc,nrc->nc", A, A)       # recomputed at the start of every sweep
        # columns whose norm fell below eps * (largest column norm) are numerically
        # zero (16x below numpy's rank threshold); they are frozen, not rotated
        zthr = (EPS * EPS) * nrm.max(axis=1, keepdims=True)
        again = np.zeros(N, dtype=bool)
        for m in range(1, 16):
            partner = lanes ^ m
            B = A[:, :, partner]                   # partner column (shuffle)
            g = np.einsum("nrc,nrc->nc", A, B)     # same value on both lanes of a pair
            nb = nrm[:, partner]
            lo = lanes < partner                   # this lane plays 'p' (lower index)
            alpha = np.where(lo, nrm, nb)
            beta = np.where(lo, nb, nrm)
            live = np.minimum(alpha, beta) > zthr
            do = live & (g * g > tol * tol * alpha * beta) & active[:, None]
            again |= (live & (g * g > EARLY * EARLY * alpha * beta)).any(axis=1)
            d = beta - alpha
            h = 2.0 * np.where(do, g, 0.5)
            t = np.where(d < 0, -1.0, 1.0) * h / (np.abs(d) + np.sqrt(d * d + h * h))
            c = 1.0 / np.sqrt(1.0 + t * t)
            s = c * t
            # lane lo: new = c*own - s*other ; lane hi: new = c*own + s*other
            sg = np.where(lo, -s, s)
            c = np.where(do, c, 1.0)
            sg = np.where(do, sg, 0.0)
            A = c[:, None, :] * A + sg[:, None, :] * B
            tg = np.where(do, t * g, 0.0)
            nrm = np.maximum(np.where(lo, nrm - tg, nrm + tg), 0.0)
        sweeps[active] += 1
        active &= again
        if not active.any():
            break
    sv = np.sqrt(np.einsum("nrc,nrc->nc", A, A))
    sv = -np.sort(-sv, axis=1)
    if return_sweeps:
        return sv, sweeps
    return sv


def rank_from_sv(sv: np.ndarray) -> np.ndarray:
    """numpy.linalg.matrix_rank default rule: count(S > S.max() * 16 * eps)."""
    return (sv > sv.max(axis=-1, keepdims=True) * 16 * EPS).sum(axis=-1)


def scores_from_sv(sv3: np.ndarray):
    """sv3: [N,3,16] descending.  Returns (scores [N,3], topo [N], rank [N,3])."""
    rank = rank_from_sv(sv3)
    minrank = np.minimum(10, rank.min(axis=1))
    keep = np.arange(16)[None, None, :] >= minrank[:, None, None]
    scores = np.sqrt((np.where(keep, sv3, 0.0) ** 2).sum(axis=2))
    return scores, scores.argmin(axis=1), rank
