#!/usr/bin/env python
"""c5-shaped flow on one GPU: source upload once, then per replicate: device bootstrap (locus
resample + within-locus shuffle + IUPAC resolution + layout) and one pass over a fresh quartet
sample.  Prints per-replicate milliseconds of the two steps."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tetrad_amd import bootstrap, synth
from tetrad_amd.engine import QuartetEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
T, S, _ = synth.CONFIGS["c3"]
tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS["c5"])
ascii_ = np.array([65, 67, 71, 84], np.uint8)
seqarr = np.where(tmparr <= 3, ascii_[np.minimum(tmparr, 3)], 78).astype(np.uint8)
rs = np.random.default_rng(0)
amb = rs.random(seqarr.shape) < 0.01
seqarr[amb] = rs.choice(np.array([82, 75, 83, 89, 87, 77], np.uint8), size=int(amb.sum()))
spans = bootstrap.get_spans(tmpmap)
dev = torch.device("cuda:0")
eng = QuartetEngine(0)
eng.set_source(seqarr, spans)
rng = np.random.default_rng(synth.CONFIG_SEEDS["c5"])
d_rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
d_rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
d_flags = torch.zeros(Q, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
tb, tr, ts = [], [], []
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Srep = bootstrap.resample_tmp_database(eng, rng)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    idx = rng.choice(synth.comb(T, 4), size=Q, replace=False)       # fresh sample per replicate (combinations.py:113)
    d_ranks = torch.from_numpy(idx.astype(np.int64)).to(dev)
    d_q = torch.zeros((Q, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.unrank_dev(d_ranks.data_ptr(), Q, d_q.data_ptr(), stream)
    eng.resolve_dev(d_q.data_ptr(), Q, True, d_rstat.data_ptr(), d_rscor.data_ptr(), d_flags.data_ptr(), stream)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    fl = d_flags.cpu().numpy()
    tb.append((t1 - t0) * 1e3); ts.append((t2 - t1) * 1e3); tr.append((t3 - t2) * 1e3)
    print(f"rep {rep}: S={Srep} bootstrap {tb[-1]:.2f} ms, host quartet sampling {ts[-1]:.1f} ms, "
          f"unrank+resolve {tr[-1]:.2f} ms, flagged rows {int((fl != 0).sum())}")
print(f"median: bootstrap {np.median(tb):.2f} ms, resolve {np.median(tr):.2f} ms per replicate of {Q} quartets")

# The same flow pipelined: the host draws of replicate k+1 (locus indices, the two seeds, the rank
# sample -- in the reference's order on the one Generator) are made while the kernels of replicate k run.
rng = np.random.default_rng(synth.CONFIG_SEEDS["c5"])
def draws():
    d = bootstrap.draw_replicate(eng.nloci, rng)
    return d, rng.choice(synth.comb(T, 4), size=Q, replace=False).astype(np.int64)
d_q = torch.zeros((Q, 4), dtype=torch.int32, device=dev)
nxt = draws()
torch.cuda.synchronize(); t0 = time.perf_counter()
for rep in range(reps):
    (lidxs, s1, s2), idx = nxt
    eng.bootstrap(lidxs, s1, s2)
    d_ranks = torch.from_numpy(idx).to(dev)
    eng.unrank_dev(d_ranks.data_ptr(), Q, d_q.data_ptr(), stream)
    eng.resolve_dev(d_q.data_ptr(), Q, True, d_rstat.data_ptr(), d_rscor.data_ptr(), d_flags.data_ptr(), stream)
    if rep + 1 < reps:
        nxt = draws()                   # host work under the GPU's
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"pipelined: {dt*1e3:.1f} ms per replicate ({Q/dt/1e6:.1f} M quartets/s including bootstrap and sampling)")
