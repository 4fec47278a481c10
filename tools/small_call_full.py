"""Full-mode calls at the reference's chunk sizes (run_inference.py:73-96), lexicographic chunks of the c2 enumeration and
random chunks of the c3 sample: the joint-histogram scan (scan_dp.hpp) switched on from different batch sizes."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine

for cfg in ("c2", "c3"):
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    q = synth.all_quartets(T) if cfg == "c2" else synth.random_quartets(T, 400_000, 4242)
    eng = QuartetEngine(0)
    eng.set_data(tmparr, tmpmap)
    for n in (1_000, 2_000, 4_000, 8_000, 16_000, 31_250, 62_500, 125_000):
        line = f"{cfg} {'lexicographic' if cfg == 'c2' else 'random'} chunk {n:7d}:"
        for dpmin in (1 << 30, 2):
            eng.set_option("dp_min_quartets", dpmin)
            eng.resolve(q[:n], False)
            reps = max(5, 400_000 // n)
            t0 = time.perf_counter()
            for i in range(reps):
                eng.resolve(q[i * n % (len(q) - n):][:n], False)
            dt = (time.perf_counter() - t0) / reps
            line += f"  {'dp' if dpmin == 2 else 'no dp'} {dt*1e3:7.3f} ms"
        print(line, flush=True)
    eng.close()
