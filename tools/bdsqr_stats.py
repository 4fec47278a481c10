#!/usr/bin/env python
"""Work statistics of tq_bdsqr_kernel on a benchmark shape: rotation steps per matrix (what a lane needs), lane-slots
issued (what its wave pays: in every sweep the wave runs for its longest block) and sweeps per matrix."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine

for cfg, Q in (("c2", 635376), ("c3", 1_000_000), ("c4", 1_000_000)):
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    q = synth.all_quartets(T) if cfg == "c2" else synth.random_quartets(T, Q, 4242)
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        for sub in (True, False):
            eng.set_option("bdsqr_stats", 1)
            eng.resolve(q, sub)
            n, steps, slots, sweeps = (int(x) for x in eng.debug_fetch("bdsqr_stats", 0))
            eng.set_option("bdsqr_stats", 0)
            print(f"{cfg} sub={sub}: {n} matrices, {steps / n:.1f} rotation steps and {sweeps / n:.1f} sweeps per matrix, "
                  f"{slots / n:.1f} lane-slots issued per matrix -> {steps / slots:.3f} of the issued slots do work")
