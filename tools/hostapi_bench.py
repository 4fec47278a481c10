"""PCIe-inclusive throughput of the host-buffer API at c3 size (warm)."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
T,S,_=synth.CONFIGS["c3"]
tmparr,tmpmap=synth.simulate_tmparr(T,S,103)
Q=1_000_000
q=synth.random_quartets(T,Q,4242)
eng=QuartetEngine(0); eng.set_data(tmparr,tmpmap)
for kv in sys.argv[1:]:                  # engine options: name=value
    k,v=kv.split("="); eng.set_option(k,int(v)); print("option",k,v)
# host-buffer API (tq_resolve): quartets H2D + kernels + rows D2H, synchronous.  A chunked
# 3-stage pipeline with pinned staging was tried and was 5-10 % SLOWER at every chunk size
# (smaller launches + the extra host copy cost more than the overlap gains), so it was dropped.
eng.resolve(q, True)
ts=[]
for _ in range(5):
    t0=time.perf_counter(); r=eng.resolve(q, True); ts.append(time.perf_counter()-t0)
print(f"tq_resolve, {Q} quartets: {min(ts)*1e3:7.2f} ms  {Q/min(ts)/1e6:6.2f} Mq/s (PCIe inclusive)")
# chunk sizes the reference's distributor produces (run_inference.py:73-96: nquartets // (breaks * ncores))
for n in (1_000, 8_000, 31_250, 125_000):
    eng.resolve(q[:n], True)
    reps = max(3, 200_000 // n)
    t0 = time.perf_counter()
    for i in range(reps):
        eng.resolve(q[i * n % (Q - n):][:n], True)
    dt = (time.perf_counter() - t0) / reps
    print(f"tq_resolve, chunk {n:7d}: {dt*1e3:7.3f} ms per call  {n/dt/1e6:6.2f} Mq/s")

# the same chunks in lexicographic order (what the reference's full mode hands out): the device sort is skipped
ql = q[np.lexsort((q[:, 3], q[:, 2], q[:, 1], q[:, 0]))]
for n in (1_000, 8_000, 31_250, 125_000):
    eng.resolve(ql[:n], True)
    reps = max(3, 200_000 // n)
    t0 = time.perf_counter()
    for i in range(reps):
        eng.resolve(ql[i * n % (Q - n):][:n], True)
    dt = (time.perf_counter() - t0) / reps
    print(f"tq_resolve, sorted chunk {n:7d}: {dt*1e3:7.3f} ms per call  {n/dt/1e6:6.2f} Mq/s")
