"""Small calls through the host-buffer API (the reference's chunk sizes, run_inference.py:73-96): wall time per call and,
under `rocprofv3 --kernel-trace --stats`, the kernels of one call.   python tools/small_call_trace.py [n] [reps]"""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T, S, _ = synth.CONFIGS["c3"]
tmparr, tmpmap = synth.simulate_tmparr(T, S, 103)
q = synth.random_quartets(T, 200_000, 4242)
eng = QuartetEngine(0); eng.set_data(tmparr, tmpmap)
for kv in sys.argv[3:]:                     # engine options, name=value
    k, v = kv.split("=")
    eng.set_option(k, int(v))
    print("option", k, v)
for sub in (True, False):
    for _ in range(5):
        eng.resolve(q[:n], sub)
    t0 = time.perf_counter()
    for i in range(reps):
        eng.resolve(q[i * n % (len(q) - n):][:n], sub)
    dt = (time.perf_counter() - t0) / reps
    print(f"tq_resolve, {n} quartets, subsample={sub}: {dt*1e3:.3f} ms per call")
