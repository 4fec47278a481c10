import sys, time
import numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tetrad_amd.engine import QuartetEngine
from oracle import oracle as orc
orc.build()
T, S = 2200, 2_000_000            # T * Sp > 2^32: the cooperative kernel's 32-bit offsets do not apply
rng = np.random.default_rng(1)
tmparr = rng.integers(0, 4, size=(T, S), dtype=np.uint8)
tmparr[:, ::7] = tmparr[0, ::7]      # many invariant sites
miss = rng.random((64, S)) < 0.1
tmparr[:64][miss] = 78
tmpmap = np.zeros((S, 2), np.uint32); tmpmap[:, 0] = np.arange(S) // 5; tmpmap[:, 1] = np.arange(S)
q = np.array([[0, 1, 2, 3], [5, 700, 1500, 2199], [63, 64, 2198, 2199], [10, 11, 12, 2100]] * 20, np.uint32)
with QuartetEngine(0) as eng:
    t0 = time.time(); eng.set_data(tmparr, tmpmap); print("set_data", round(time.time() - t0, 1), "s")
    for sub in (True, False):
        t0 = time.time(); rstat, rscor, flags = eng.resolve(q, sub); dt = time.time() - t0
        _, o_rstat, o_rscor = orc.new_infer_resolved_quartets(tmparr, tmpmap, q[:4], sub)
        assert np.array_equal(rstat[:4], o_rstat), (rstat[:4], o_rstat)
        assert np.allclose(rscor[:4], o_rscor, rtol=1e-6)
        assert np.array_equal(rstat[:4], rstat[4:8])
        print("sub", sub, "ok", rstat[:4, 1], round(dt * 1e3, 1), "ms for", len(q), "quartets")
print("big-shape check ok")
