import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
T,S,_=synth.CONFIGS["c3"]
tmparr,tmpmap=synth.simulate_tmparr(T,S,103)
q=synth.random_quartets(T,200_000,4242)
eng=QuartetEngine(0); eng.set_data(tmparr,tmpmap)
for n in (500,1000,1500,2000,3000,4000,6000):
    line=f"{n:6d}:"
    for wgmin in (64, 1<<20):
        eng.set_option("wg_min_quartets", wgmin)
        for sub in (True, False):
            eng.resolve(q[:n], sub)
            reps=max(20, 100_000//n)
            t0=time.perf_counter()
            for i in range(reps): eng.resolve(q[i*n%(len(q)-n):][:n], sub)
            dt=(time.perf_counter()-t0)/reps
            line+=f"  {'coop' if wgmin==64 else 'one-wave'} {'sub' if sub else 'full'} {dt*1e3:.3f}"
    print(line, flush=True)
