#!/usr/bin/env python
"""What is ordering the matrices into waves worth for tq_bdsqr_kernel?  (review of round 3: 61.6 of 300 issued lane-slots
per matrix are idle INSIDE sweeps -- in every sweep a wave runs for the longest block among its 64 lanes.)

The bidiagonals of one singular-value chunk of the c3 benchmark are fetched (tq_debug_fetch), the QR kernel is run alone on
them (tq_debug_bdsqr) in the engine's order, then in orders a predictor available after bidiagonalisation could produce, and
in the order of PERFECT knowledge (sorted by the rotation steps / sweeps the kernel itself counted): the upper bound of any
predictor.  Singular values must be the same set per matrix in every order."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
T, S, _ = synth.CONFIGS[cfg]
tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
nq = 200_000
q = synth.random_quartets(T, nq, 4242) if cfg != "c2" else synth.all_quartets(T)[:nq]
with QuartetEngine(0) as eng:
    eng.set_data(tmparr, tmpmap)
    for sub in (True, False):
        eng.set_option("svd_streams", 1)
        eng.set_option("svd_chunk", nq)
        eng.resolve(q, sub)
        de = eng.debug_fetch("de", nq)                       # f64 [3 nq, 32]: d[16], e[16]
        n = de.shape[0]
        sv0, steps, sweeps, ms0 = eng.debug_bdsqr(de)
        d, e = np.abs(de[:, :16]), np.abs(de[:, 16:])
        anorm = (d + e).max(axis=1)
        print(f"{cfg} sub={sub}: {n} matrices, {steps.mean():.1f} rotation steps (sd {steps.std():.1f}), {sweeps.mean():.1f} sweeps "
              f"(sd {sweeps.std():.1f}) per matrix; engine order {ms0:.3f} ms per launch")
        orders = {
            "perfect knowledge: sorted by rotation steps": np.argsort(steps, kind="stable"),
            "perfect knowledge: sorted by (sweeps, steps)": np.lexsort((steps, sweeps)),
            "predictor: smallest |d| / anorm": np.argsort(d.min(axis=1) / anorm, kind="stable"),
            "predictor: number of |e| < 1e-3 anorm": np.argsort((e[:, 1:] < 1e-3 * anorm[:, None]).sum(axis=1), kind="stable"),
            "predictor: sum |e| / sum |d|": np.argsort(e.sum(axis=1) / d.sum(axis=1), kind="stable"),
            "predictor: d[15] / anorm (bottom pivot)": np.argsort(d[:, 15] / anorm, kind="stable"),
            "predictor: condition estimate max|d| / min|d|": np.argsort(d.max(axis=1) / np.maximum(d.min(axis=1), 1e-300), kind="stable"),
            "random permutation": np.random.default_rng(1).permutation(n),
        }
        for name, perm in orders.items():
            sv, st, sw, ms = eng.debug_bdsqr(de[perm])
            same = np.array_equal(np.sort(sv, axis=1), np.sort(sv0[perm], axis=1))
            r = np.corrcoef(np.arange(n), steps[perm])[0, 1]
            print(f"  {name:55s} {ms:.3f} ms ({ms / ms0 - 1:+.1%})  values identical: {same}  corr(order, steps) {r:+.2f}")
        eng.set_option("svd_streams", 0)
        eng.set_option("svd_chunk", 0)
