import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from tetrad_amd import synth
from oracle import oracle as orc
import dqds_model as dm
orc.build()
tmparr, tmpmap, q = synth.make_config("c3", Q=400)
_, rstat, rscor, dbg = orc.new_infer_resolved_quartets(tmparr, tmpmap, q[:300], True, debug=True)
mats = dbg["cmats"].reshape(-1,16,16).astype(float)
d, e = dm.bidiagonalize(mats)
ref = np.linalg.svd(mats, compute_uv=False)
EPS = np.finfo(float).eps
TOL2 = (100*EPS)**2

def redo_dmin(dh, l, k):
    if dh is None or k < l: return None, 0
    seg = dh[l:k+1]
    if not np.isfinite(seg).all(): return None, 0
    j = int(np.argmin(seg)); return float(seg[j]), l + j

def dqds_one(d, e, strategy):
    n = 16
    q = d*d
    ee = np.zeros(n); ee[:n-1] = e*e
    lam = np.zeros(n)
    steps = 0; sweeps = 0; fails = 0
    # blocks stack: (l, k, sigma)
    stack = [(0, n-1, 0.0)]
    while stack:
        l, k, sigma = stack.pop()
        dmin = None; told = 0.0; fail_streak = 0; dhist = None; dmin_at = 0
        dn = dn1 = None
        while k >= l:
            if k == l:
                lam[k] = q[k] + sigma; k -= 1; break
            # bottom deflation
            if ee[k-1] <= TOL2 * (sigma + q[k]) or ee[k-1] <= TOL2*q[k-1]*0 + 0 and False:
                lam[k] = q[k] + sigma; k -= 1; dmin, dmin_at = redo_dmin(dhist, l, k); continue
            # 2x2 at bottom if e[k-2] negligible (or k-1==l)
            if k-1 == l or ee[k-2] <= TOL2 * (sigma + q[k-1]):
                # eigenvalues of [[q[k-1], 1],[..]] qd 2x2: matrix L U with q1=q[k-1], e1=ee[k-1], q2=q[k]
                q1, e1, q2 = q[k-1], ee[k-1], q[k]
                # T = [[q1, sqrt(q1 e1)],[sqrt(q1 e1), q2+e1]]  (B^T B of 2x2 bidiagonal with d1^2=q1, e^2=e1, d2^2=q2)
                tr = q1 + q2 + e1; det = q1*q2
                disc = np.sqrt(max(0.0, (q1 - q2 - e1)**2 + 4*q1*e1)) if True else 0
                # stable: larger root
                big = 0.5*(tr + np.sqrt(max(0.0,(q1+e1-q2)**2 + 4*e1*q2)))
                small = det/big if big > 0 else 0.0
                lam[k-1] = big + sigma; lam[k] = small + sigma
                k -= 2; dmin, dmin_at = redo_dmin(dhist, l, k); continue
            # split search
            split = None
            for i in range(k-2, l-1, -1):
                if ee[i] <= TOL2 * (sigma + q[i+1]) and ee[i] <= TOL2*(sigma+q[i]):
                    split = i; break
            if split is not None:
                stack.append((l, split, sigma))
                l = split + 1
                dmin=None
                continue
            # shift
            if strategy == "zero":
                tau = 0.0
            else:
                if dmin is None or dmin <= 0:
                    # first sweep of a block: Gershgorin-ish lower bound: min over i of q_i + e_{i-1} - sqrt(q_i e_i) - sqrt(q_{i-1} e_{i-1})
                    tau = 0.0
                else:
                    if strategy == "quarter":
                        tau = 0.25*dmin
                    else:
                        # dlasq4-like: cases by where dmin occurred
                        b1 = ee[k-1]/q[k-1] if q[k-1] > 0 else 1.0
                        b2 = b1
                        if k-2 >= l and q[k-2] > 0:
                            b2 = b1 + b1*ee[k-2]/q[k-2]
                        if dmin_at >= k-1:      # dmin at the bottom: converging there
                            # Rayleigh-like: tau = dmin*(1 - sqrt(b2))/(1+b2) if b2<1
                            if b2 < 0.5:
                                tau = dmin*(1 - np.sqrt(b2)) / (1 + b2)
                            else:
                                tau = FB*dmin
                        else:
                            tau = FO*dmin
                        if fail_streak:
                            tau *= 0.25**fail_streak
            # sweep
            while True:
                qq = np.empty(n); e2 = np.empty(n)
                dd = q[l] - tau
                ok = dd >= 0
                dmin_new = dd; at = l; dh = np.full(n, np.inf); dh[l] = dd
                for i in range(l, k):
                    qq[i] = dd + ee[i]
                    if qq[i] <= 0: ok = False; break
                    t = q[i+1]/qq[i]
                    e2[i] = ee[i]*t
                    dd = dd*t - tau
                    dh[i+1] = dd
                    if dd < dmin_new: dmin_new = dd; at = i+1
                    if dd < 0: ok = False; break
                steps += (k-l); sweeps += 1
                if ok:
                    qq[k] = dd
                    q[l:k+1] = qq[l:k+1]; ee[l:k] = e2[l:k]
                    sigma += tau; dmin = dmin_new; dmin_at = at; fail_streak = 0; dhist = dh
                    break
                else:
                    fails += 1; fail_streak += 1
                    if fail_streak >= 3: tau = 0.0
                    else: tau *= 0.25
    return np.sqrt(np.maximum(lam,0)), steps, sweeps, fails

for FB,FO in ((0.25,0.25),(0.5,0.5),(0.9,0.9)):
  for strat in ("lasq",):
    tot=0; sw=0; fl=0; worst=0
    for i in range(0,len(d),3):
        sv, st, s_, f_ = dqds_one(d[i].copy(), e[i].copy(), strat)
        sv = -np.sort(-sv)
        err = np.max(np.abs(sv-ref[i])/max(ref[i][0],1e-300))
        worst=max(worst,err); tot+=st*3; sw+=s_*3; fl+=f_*3
    print(FB,FO,strat, "steps/matrix", tot/len(d), "sweeps", sw/len(d), "fails", fl/len(d), "worst abs err/smax", worst)
