#!/usr/bin/env python
"""Interleaved A/B timing of kernel variants in ONE process (HIP-event kernel time).

usage: python tools/sweep.py [--config c3] [--quartets 200000] [--rounds 3]
Variants are (mode, nrep, waves_per_cu, phases); phases 1/2 are the scan-only / SVD-only
diagnostic builds.  Prints one line per variant: median and min kernel ms, quartets/s.
"""
import argparse
import itertools
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--quartets", type=int, default=200_000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--nreps", default="1,2,4,8,16,32")
    ap.add_argument("--wpcs", default="0")
    ap.add_argument("--phases", default="3,1,2")
    ap.add_argument("--modes", default="sub,full")
    args = ap.parse_args()
    import torch
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine

    T, S, _ = synth.CONFIGS[args.config]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[args.config])
    Q = args.quartets
    quartets = synth.random_quartets(T, Q, 4242)
    dev = torch.device("cuda:0")
    eng = QuartetEngine(0)
    eng.set_data(tmparr, tmpmap)
    d_q = torch.from_numpy(quartets.astype(np.int32)).to(dev)
    d_rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
    d_rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    variants = list(itertools.product(
        args.modes.split(","), [int(x) for x in args.nreps.split(",")],
        [int(x) for x in args.wpcs.split(",")], [int(x) for x in args.phases.split(",")]))
    times = {v: [] for v in variants}
    eng.timing_enable(True)
    for rnd in range(args.rounds + 1):
        for v in variants:
            mode, nrep, wpc, ph = v
            eng.set_option("nrep", nrep)
            eng.set_option("waves_per_cu", wpc)
            eng.set_option("phases", ph)
            eng.resolve_dev(d_q.data_ptr(), Q, mode == "sub", d_rstat.data_ptr(), d_rscor.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            ms, n = eng.timing_read()
            if rnd:                     # round 0 is warm-up
                times[v].append(ms)
    print(f"# {args.config} T={T} S={S} Q={Q} rounds={args.rounds}")
    print("mode nrep wpc phases  med_ms   min_ms   Mq/s(min)")
    for v in variants:
        t = np.array(times[v])
        print(f"{v[0]:4s} {v[1]:4d} {v[2]:3d} {v[3]:6d} {np.median(t):8.3f} {t.min():8.3f} {Q / t.min() / 1e3:9.2f}")


if __name__ == "__main__":
    main()
