#!/usr/bin/env python
"""c5 end to end INCLUDING the supertree step: bootstrap replicates (device resample, 1e6-quartet sample, resolve) and one
clean-room quartet-MaxCut tree per replicate on a host thread pool.  Prints replicates/s and how many of the generating
tree's bipartitions the replicate trees carry."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from test_qmc_tree import _bipartitions_from_children, _bipartitions_from_newick
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
from tetrad_amd.replicates import bootstrap_trees

nboots = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seqarr, maparr, spans = synth.make_c5_source()
T = seqarr.shape[0]
children, root = synth.random_tree_children(T, np.random.default_rng(synth.CONFIG_SEEDS["c5"]))
truth = _bipartitions_from_children(children, root, T)
for sampler in ("host", "device"):
    with QuartetEngine(0) as eng:
        bootstrap_trees(eng, seqarr, spans, 1_000_000, 2, weights=1, seed=1, sampler=sampler, workers=workers)   # warm-up
        t0 = time.perf_counter()
        trees = bootstrap_trees(eng, seqarr, spans, 1_000_000, nboots, weights=1, seed=2, sampler=sampler, workers=workers)
        dt = time.perf_counter() - t0
    found = [len(_bipartitions_from_newick(t, T) & truth) for t in trees]
    print(f"sampler={sampler}: {nboots} replicates x 1e6 quartets with trees in {dt:.2f} s = {nboots / dt:.1f} replicates/s "
          f"({nboots * 1e6 / dt / 1e6:.1f} M quartets/s end to end, {workers} tree threads); true bipartitions per tree "
          f"{min(found)}-{max(found)} of {len(truth)}")
