#!/usr/bin/env python
"""A/B of the two-stream overlapped pipeline (tq_set_option "overlap") vs the sequential one.
Wall time around K back-to-back resolve calls, one process, interleaved rounds."""
import sys, time, itertools
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine

T, S, _ = synth.CONFIGS["c3"]
tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS["c3"])
Q = 1_000_000
quartets = synth.random_quartets(T, Q, 4242)
dev = torch.device("cuda:0")
eng = QuartetEngine(0)
eng.set_data(tmparr, tmpmap)
d_q = torch.from_numpy(quartets.astype(np.int32)).to(dev)
d_rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
d_rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ref = None
variants = [(0, 1, 6)] + [(sub, wg, 6) for sub in (65536, 131072, 262144, 524288) for wg in (0, 2, 3, 4)]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        sub, wg, sw = v
        eng.set_option("overlap", sub)
        eng.set_option("ov_scan_wgs", wg)
        eng.set_option("ov_svd_waves", sw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            eng.resolve_dev(d_q.data_ptr(), Q, True, d_rstat.data_ptr(), d_rscor.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        if rnd:
            res[v].append(dt)
        r = (d_rstat.cpu().numpy(), d_rscor.cpu().numpy())
        if ref is None:
            ref = r
        else:
            assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]), v
print("overlap_sub scan_wgs svd_waves   ms/1e6   Mq/s")
for v in variants:
    t = min(res[v])
    print(f"{v[0]:10d} {v[1]:8d} {v[2]:9d} {t*1e3:8.2f} {Q/t/1e6:7.2f}")
