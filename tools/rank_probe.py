import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
tmparr, tmpmap = synth.simulate_tmparr(30, 400, seed=77, p=0.02, missing=0.35)
q = synth.all_quartets(30)
res = {}
with QuartetEngine(0) as eng:
    eng.set_data(tmparr, tmpmap)
    for name, m in (("hqr", 1), ("jacobi", 0)):
        eng.set_option("svd_method", m)
        res[name] = eng.resolve(q, True, debug=True)
        if name == "hqr":
            de = eng.debug_fetch("de", len(q))
            svraw = eng.debug_fetch("sv", len(q))
d1, d0 = res["hqr"][3], res["jacobi"][3]
bad = np.argwhere(d1["ranks"] != d0["ranks"])
eps = np.finfo(float).eps
for qi, t in bad:
    M = d0["cmats"][qi, t].astype(float)
    sv = np.linalg.svd(M)[1]
    print("quartet", qi, "t", t, "rank hqr", d1["ranks"][qi, t], "jacobi", d0["ranks"][qi, t], "numpy", np.linalg.matrix_rank(M))
    print("  thr/smax = %.3e" % (16 * eps))
    print("  numpy  sv/smax:", np.array2string(sv / sv[0], precision=3))
    print("  hqr    sv/smax:", np.array2string(d1["svds"][qi, t] / sv[0], precision=3))
    print("  jacobi sv/smax:", np.array2string(d0["svds"][qi, t] / sv[0], precision=3))
    print("  device d:", np.array2string(de[3 * qi + t, :16], precision=4))
    print("  device e:", np.array2string(de[3 * qi + t, 16:], precision=4))
    print("  device sv raw:", np.array2string(svraw[3 * qi + t], precision=4))
# overall agreement with numpy ranks on a sample
rng = np.random.default_rng(0)
idx = rng.choice(len(q), 3000, replace=False)
mm = {"hqr": 0, "jacobi": 0}
for qi in idx:
    for t in range(3):
        r = np.linalg.matrix_rank(d0["cmats"][qi, t].astype(float))
        mm["hqr"] += int(r != d1["ranks"][qi, t]); mm["jacobi"] += int(r != d0["ranks"][qi, t])
print("rank mismatches vs numpy on 9000 matrices:", mm)
