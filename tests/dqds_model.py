"""NumPy model of the device singular-value path "bidiagonalise, then dqds" (tetrad_amd/csrc/hqr.hpp),
vectorised over matrices exactly the way the kernel is vectorised over lanes: every matrix runs the
same fixed-length sweeps, splits and deflation are handled by selects, never by index logic.

Replaces np.linalg.svd(mat)[1] of tetrad/src/resolve_quartets.py:242 (LAPACK dgesdd, which for
singular values only ends in the same dqds family, dlasq1).  Test infrastructure: used by
tests/test_dqds_model.py on CPU to pin the algorithm (iteration counts, accuracy, rank parity with
numpy) before the HIP kernel is trusted.
"""
from __future__ import annotations

import numpy as np

EPS = np.finfo(np.float64).eps
TOL2 = (4.0 * EPS) ** 2          # e[j] <= TOL2 * (sigma + q[j]) is flushed to zero (relative criterion)
MAX_SWEEPS = 200


def bidiagonalize(mats: np.ndarray):
    """Householder bidiagonalisation of [N,16,16] -> (d[N,16], e[N,15]) (upper bidiagonal), with the
    kernel's rule that columns / rows whose squared norm is below 1e-40*||M||_F^2 are exactly zero."""
    a = np.array(mats, dtype=np.float64, copy=True)
    N, n, _ = a.shape
    thr2 = (a * a).sum(axis=(1, 2)) * 1e-40
    d = np.zeros((N, n))
    e = np.zeros((N, n - 1))
    for k in range(n):
        x = a[:, k:, k]
        n2 = (x * x).sum(axis=1)
        live = n2 > thr2
        nrm = np.sqrt(n2)
        alpha = np.where(live, np.where(x[:, 0] < 0, nrm, -nrm), 0.0)
        v = x.copy()
        v[:, 0] -= alpha
        den = n2 - alpha * x[:, 0]
        beta = np.where(live, 1.0 / np.where(live, den, 1.0), 0.0)
        d[:, k] = alpha
        w = np.einsum("nr,nrc->nc", v, a[:, k:, k + 1:]) * beta[:, None]
        a[:, k:, k + 1:] -= v[:, :, None] * w[:, None, :]
        a[:, k:, k] = 0.0
        if k <= n - 3:
            y = a[:, k, k + 1:]
            n2 = (y * y).sum(axis=1)
            live = n2 > thr2
            nrm = np.sqrt(n2)
            alpha = np.where(live, np.where(y[:, 0] < 0, nrm, -nrm), 0.0)
            v = y.copy()
            v[:, 0] -= alpha
            den = n2 - alpha * y[:, 0]
            beta = np.where(live, 1.0 / np.where(live, den, 1.0), 0.0)
            e[:, k] = alpha
            w = np.einsum("nc,nrc->nr", v, a[:, k + 1:, k + 1:]) * beta[:, None]
            a[:, k + 1:, k + 1:] -= w[:, :, None] * v[:, None, :]
            a[:, k, k + 1:] = 0.0
        elif k == n - 2:
            e[:, k] = a[:, k, k + 1]
    return d, e


def dqds_singular_values(d: np.ndarray, e: np.ndarray, shift: str = "laguerre", max_sweeps: int = MAX_SWEEPS,
                         return_sweeps: bool = False):
    """Singular values (unsorted) of N upper-bidiagonal matrices (d[N,n], e[N,n-1]) by dqds sweeps over the
    whole matrix.  shift = "newton" | "laguerre" (guaranteed lower bounds of the smallest live eigenvalue
    from sums accumulated inside the sweep) | "zero"."""
    N, n = d.shape
    with np.errstate(all="ignore"):
        # scale so that squares cannot overflow / underflow (counts are <= 2^32 anyway)
        q = d * d
        ee = np.zeros((N, n))
        ee[:, : n - 1] = e * e                       # ee[:, n-1] = 0 closes the last block
        sigma = np.zeros(N)
        tau = np.zeros(N)
        dead = np.zeros((N, n), dtype=bool)
        # flush and initial dead detection
        neg = ee <= TOL2 * (sigma[:, None] + q)
        ee = np.where(neg, 0.0, ee)
        sweeps = np.zeros(N, dtype=np.int64)
        for it in range(max_sweeps + 1):
            z = ee == 0.0
            zprev = np.concatenate([np.ones((N, 1), bool), z[:, :-1]], axis=1)
            newdead = z & zprev & ~dead
            q = np.where(newdead, q + sigma[:, None], q)          # frozen at its absolute value
            dead |= newdead
            alive = ~dead.all(axis=1)
            if not alive.any() or it == max_sweeps:
                break
            sweeps += alive
            # ---- one sweep (all matrices, all 16 positions) ----
            tau_eff = np.where(dead, 0.0, tau[:, None])
            sigma_new = sigma + tau
            dcur = q[:, 0] - tau_eff[:, 0]
            c = np.zeros(N)
            w = np.zeros(N)
            S1 = np.zeros(N)
            S2 = np.zeros(N)
            m = (~dead).sum(axis=1).astype(np.float64)
            eprev = np.zeros(N)
            qn = np.empty_like(q)
            en = np.zeros_like(ee)
            for j in range(n):
                ej = ee[:, j]
                split = ej == 0.0
                dcur = np.where(dead[:, j], dcur, np.maximum(dcur, 0.0))
                qq = dcur + ej
                rinv = 1.0 / qq
                # Newton / Laguerre sums for the matrix produced by this sweep (positions that stay live)
                c2 = c * c
                rho = eprev * rinv
                w = rho * (c2 + w)
                c = (eprev * c + 1.0) * rinv
                live_j = ~dead[:, j]
                S1 = S1 + np.where(live_j, c, 0.0)
                S2 = S2 + np.where(live_j, c * c + 2.0 * w, 0.0)
                qn[:, j] = qq
                if j < n - 1:
                    t = q[:, j + 1] * rinv
                    enew = ej * t
                    enew = np.where(split | (enew <= TOL2 * (sigma_new + qq)), 0.0, enew)
                    dn = dcur * t - tau
                    fresh = q[:, j + 1] - tau_eff[:, j + 1]
                    en[:, j] = enew
                    dcur = np.where(split, fresh, dn)
                    # a block ends here: restart the recurrences
                    c = np.where(split, 0.0, c)
                    w = np.where(split, 0.0, w)
                    eprev = np.where(split, 0.0, enew)
            q, ee, sigma = qn, en, sigma_new
            if shift == "zero":
                tau = np.zeros(N)
            elif shift == "newton":
                tau = np.where(np.isfinite(S1) & (S1 > 0), 1.0 / S1, 0.0)
            else:
                disc = (m - 1.0) * (m * S2 - S1 * S1)
                den = S1 + np.sqrt(np.maximum(disc, 0.0))
                tau = np.where(np.isfinite(den) & (den > 0), m / den, 0.0)
                tnewton = np.where(np.isfinite(S1) & (S1 > 0), 1.0 / S1, 0.0)
                tau = np.where(np.isfinite(tau), np.maximum(tau, tnewton), tnewton)
            tau = tau * (1.0 - 64.0 * EPS)
        lam = np.where(dead, q, q + sigma[:, None])
        sv = np.sqrt(np.maximum(lam, 0.0))
    if return_sweeps:
        return sv, sweeps
    return sv


def singular_values(mats: np.ndarray, **kw):
    d, e = bidiagonalize(mats)
    return dqds_singular_values(d, e, **kw)
