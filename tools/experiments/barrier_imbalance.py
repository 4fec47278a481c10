"""What the per-step workgroup barrier of the cooperative scan costs: the same 1e6-quartet c3 batch scanned (a) as it is,
(b) with every quartet four times in a row -- 250 000 distinct quartets, so the four waves of a workgroup walk the same counted
sites and reach every barrier together.  Scan kernel time per 1e6 quartets, subsample mode."""
import sys
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine

T, S, _ = synth.CONFIGS["c3"]
tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS["c3"])
q = synth.random_quartets(T, 1_000_000, 1103)
q = q[np.lexsort((q[:, 3], q[:, 2], q[:, 1], q[:, 0]))]          # sorted on arrival: the order is then the array order
q4 = np.repeat(q[::4], 4, axis=0)
for f4 in (0, 1):
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        eng.set_option("scan_f4", f4)
        for name, qq in (("as it is", q), ("each quartet x 4", q4)):
            eng.resolve(qq, True)
            eng.timing_enable(True)
            for _ in range(5):
                eng.resolve(qq, True)
            k, n = eng.timing_read_kernels()
            eng.timing_enable(False)
            print(f"scan_f4={f4} {name:18s}: scan {k['scan'] / n:.3f} ms", flush=True)
