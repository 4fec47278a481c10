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

for cfg, Q in (("c3", 1_000_000),):
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    q = synth.all_quartets(T) if cfg == "c2" else synth.random_quartets(T, Q, 4242)
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        for sub in (True, False):
            eng.set_option("bdsqr_stats", 1)
            eng.resolve(q, sub)
            n, steps, slots, sweeps, tail = (int(x) for x in eng.debug_fetch("bdsqr_stats", 0)[:5])
            eng.set_option("bdsqr_stats", 0)
            print(f"{cfg} sub={sub}: {n} matrices, {steps / n:.1f} rotation steps and {sweeps / n:.1f} sweeps per matrix, "
                  f"{slots / n:.1f} lane-slots issued per matrix -> {steps / slots:.3f} of the issued slots do work; "
                  f"{tail / n:.1f} of the idle slots per matrix come after the lane's matrix is done (a refill could use them), "
                  f"{(slots - steps - tail) / n:.1f} inside sweeps (shorter block than the wave's longest)")
