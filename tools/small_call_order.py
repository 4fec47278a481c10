import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
T, S, _ = synth.CONFIGS["c3"]
tmparr, tmpmap = synth.simulate_tmparr(T, S, 103)
Q = 1_000_000
q = synth.random_quartets(T, Q, 4242)
eng = QuartetEngine(0); eng.set_data(tmparr, tmpmap)
for n in (1000, 2000, 4000, 8000, 16000, 31250, 62500, 125000):
    row = []
    for order in (1, 0):
        eng.set_option("order", order)
        eng.resolve(q[:n], True)
        reps = max(5, 400_000 // n)
        t0 = time.perf_counter()
        for i in range(reps):
            eng.resolve(q[i * n % (Q - n):][:n], True)
        row.append((time.perf_counter() - t0) / reps * 1e3)
    print(f"chunk {n:7d}: sorted order {row[0]:7.3f} ms   natural order {row[1]:7.3f} ms")
