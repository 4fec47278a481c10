"""Model of tq_bdsqr_kernel's QR iteration (hqr.hpp) on real c3 bidiagonals: in which ORDER do the singular values deflate,
and how many rotation steps does it take until the m smallest are known (checked by a Sturm count on the rest)?
The scores need the 16 - minrank smallest values only (resolve_quartets.py:246-248: minrank = min(10, rank.min()))."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from tetrad_amd import synth
from oracle import oracle as orc
import dqds_model as dm

EPS = np.finfo(float).eps


def qr_deflation_trace(d, e):
    """Golub-Kahan implicit-shift QR as the kernel does it (w = diagonal, e[i] couples i-1, i; e[0] unused).
    Returns list of (value, steps_so_far) in deflation order and total steps."""
    w = d.astype(float).copy()
    ee = np.zeros(16); ee[1:] = e
    anorm = np.max(np.abs(w) + np.abs(ee))
    tiny = anorm * 0.5 * EPS
    k = 15; its = 0; steps = 0; out = []; sweeps = 0
    while True:
        while True:
            l = k
            cancel = False
            while l > 0:
                if abs(ee[l]) <= tiny: break
                if abs(w[l - 1]) <= tiny: cancel = True; break
                l -= 1
            if cancel:
                cc, ss = 0.0, 1.0
                for i in range(l, k + 1):
                    f = ss * ee[i]; ee[i] = cc * ee[i]
                    if abs(f) <= tiny: break
                    g = w[i]; h = np.hypot(f, g); w[i] = h; cc = g / h; ss = -f / h
            if l != k:
                break
            out.append((abs(w[k]), steps)); k -= 1; its = 0
            if k < 0: return out, steps, sweeps
        its += 1; sweeps += 1
        nm = k - 1
        x = w[l]; y = w[nm]; g = ee[nm]; h = ee[k]; z = w[k]
        f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y)
        g = np.hypot(f, 1.0)
        f = ((x - z) * (x + z) + h * (y / (f + np.copysign(g, f)) - h)) / x
        cc = ss = 1.0
        for jj in range(l, nm + 1):
            i = jj + 1
            g = ee[i]; y = w[i]
            h = ss * g; g = cc * g
            zz = np.hypot(f, h); ee[jj] = zz
            cc = f / zz if zz else 0.0; ss = h / zz if zz else 0.0
            f = x * cc + g * ss; g = g * cc - x * ss; h = y * ss; y *= cc
            zz = np.hypot(f, h); w[jj] = zz
            if zz: cc = f / zz; ss = h / zz
            f = cc * g + ss * y; x = cc * y - ss * g
            steps += 1
        ee[l] = 0.0; ee[k] = f; w[k] = x


tmparr, tmpmap, q = synth.make_config("c3", Q=300)
for sub in (True, False):
    _, rstat, rscor, dbg = orc.new_infer_resolved_quartets(tmparr, tmpmap, q[:200], sub, debug=True)
    mats = dbg["cmats"].reshape(-1, 16, 16).astype(float)
    d, e = dm.bidiagonalize(mats)
    ref = np.linalg.svd(mats, compute_uv=False)
    tot = []; until6 = []; until6_checked = []; inorder = 0; pos_of_smallest6 = []
    for i in range(len(mats)):
        tr, steps, sweeps = qr_deflation_trace(d[i], e[i])
        vals = np.array([v for v, _ in tr]); at = np.array([s for _, s in tr])
        assert np.allclose(np.sort(vals)[::-1], ref[i], rtol=1e-9, atol=1e-9 * ref[i, 0])
        tot.append(steps)
        # after how many deflations are the 6 smallest all known?  (rank of the 6 smallest in deflation order)
        order = np.argsort(vals)            # indices (deflation positions) of ascending values
        need = order[:6].max()              # deflation position by which all six smallest have appeared
        pos_of_smallest6.append(need + 1)
        until6.append(at[need])
        inorder += int(need == 5)
    tot = np.array(tot); until6 = np.array(until6); pos = np.array(pos_of_smallest6)
    print(f"sub={sub}: {len(mats)} matrices; total rotation steps {tot.mean():.1f}; steps until the six smallest have all deflated "
          f"{until6.mean():.1f} ({until6.mean() / tot.mean():.2f} of the total); they are the first six to deflate in "
          f"{inorder / len(mats):.2%} of the matrices; deflations needed: mean {pos.mean():.2f}, max {pos.max()}, "
          f"histogram {np.bincount(pos)[6:].tolist()}")


# ---- wave-level cost: 64 lanes in lockstep, every sweep costs the longest block among the lanes that sweep ----
def sweep_trace(d, e, need=None):
    """Block length of every sweep of one matrix; with `need` = m the iteration stops as soon as the m smallest singular
    values are known: at least m have deflated and the still undeflated leading block has no singular value below the
    m-th smallest deflated one (Sturm count on the block's B^T B; charged as 3 rotation steps)."""
    w = d.astype(float).copy()
    ee = np.zeros(16); ee[1:] = e
    anorm = np.max(np.abs(w) + np.abs(ee))
    tiny = anorm * 0.5 * EPS
    k = 15; lens = []; done = []
    while True:
        while True:
            l = k
            while l > 0 and abs(ee[l]) > tiny and abs(w[l - 1]) > tiny:
                l -= 1
            if l != k:
                break
            done.append(abs(w[k])); k -= 1
            if k < 0: return lens
            if need is not None and len(done) >= need:
                t = np.sort(done)[need - 1]
                # Sturm count: number of singular values of the leading (k+1) x (k+1) block below t
                cnt = 0; qv = w[0] ** 2 - t * t
                cnt += qv < 0
                for i in range(1, k + 1):
                    qv = w[i] ** 2 + ee[i] ** 2 - t * t - (ee[i] ** 2) * (w[i - 1] ** 2) / (qv if qv != 0 else 1e-300)
                    cnt += qv < 0
                lens.append(3)
                if cnt == 0: return lens
        nm = k - 1
        x = w[l]; y = w[nm]; g = ee[nm]; h = ee[k]; z = w[k]
        f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y)
        g = np.hypot(f, 1.0)
        f = ((x - z) * (x + z) + h * (y / (f + np.copysign(g, f)) - h)) / x
        cc = ss = 1.0
        for jj in range(l, nm + 1):
            i = jj + 1
            g = ee[i]; y = w[i]
            h = ss * g; g = cc * g
            zz = np.hypot(f, h); ee[jj] = zz
            cc = f / zz if zz else 0.0; ss = h / zz if zz else 0.0
            f = x * cc + g * ss; g = g * cc - x * ss; h = y * ss; y *= cc
            zz = np.hypot(f, h); w[jj] = zz
            if zz: cc = f / zz; ss = h / zz
            f = cc * g + ss * y; x = cc * y - ss * g
        ee[l] = 0.0; ee[k] = f; w[k] = x
        lens.append(nm - l + 1)


def wave_cost(traces):
    n = max(len(t) for t in traces)
    return sum(max((t[i] if i < len(t) else 0) for t in traces) for i in range(n))


_, rstat, rscor, dbg = orc.new_infer_resolved_quartets(tmparr, tmpmap, q[:256], True, debug=True)
mats = dbg["cmats"].reshape(-1, 16, 16).astype(float)
d, e = dm.bidiagonalize(mats)
full = [sweep_trace(d[i], e[i]) for i in range(len(mats))]
part = [sweep_trace(d[i], e[i], need=6) for i in range(len(mats))]
for name, tr in (("full spectrum", full), ("six smallest", part)):
    lane = np.mean([sum(t) for t in tr])
    waves = [wave_cost(tr[i:i + 64]) for i in range(0, len(tr), 64)]
    print(f"{name}: {lane:.1f} steps per matrix, wave cost {np.mean(waves):.1f} lane-slots per matrix-slot "
          f"(sweeps per wave {np.mean([max(len(t) for t in tr[i:i+64]) for i in range(0, len(tr), 64)]):.1f})")
