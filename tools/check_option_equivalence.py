import sys, numpy as np
sys.path.insert(0, '.')
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
opt = sys.argv[1:]
tmparr, tmpmap, q = synth.make_config("c3", Q=200000)
with QuartetEngine(0) as e:
    e.set_data(tmparr, tmpmap)
    for sub in (True, False):
        a = e.resolve(q, sub)
        for kv in opt:
            k, v = kv.split("="); e.set_option(k, int(v))
        b = e.resolve(q, sub)
        for kv in opt:
            k, v = kv.split("="); e.set_option(k, 0)
        print("sub", sub, "equal:", all(np.array_equal(x, y) for x, y in zip(a, b)), "nsnps mean", a[0][:,1].mean())
