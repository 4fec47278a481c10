#!/usr/bin/env python
"""Interleaved A/B timing of kernel variants in ONE process (HIP-event kernel time).

usage: python tools/sweep.py [--config c3] [--quartets 200000] [--rounds 3]
Variants are (mode, nrep, waves_per_cu).  Prints one line per variant: min total / scan-kernel /
SVD-kernel milliseconds (tq_timing_read_split) and quartets/s.
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
    ap.add_argument("--modes", default="sub,full")
    ap.add_argument("--methods", default="1")
    ap.add_argument("--orders", default="1")
    ap.add_argument("--svds", default="1", help="svd_method list: 0 Jacobi, 1 HQR")
    ap.add_argument("--wgs", default="4", help="scan_wg list: 1 one wave per quartet; 2/4/8/16 waves per cooperative workgroup")
    ap.add_argument("--svdwpcs", default="0", help="svd_wpc list: blocks per CU of the bidiag/bdsqr grids (0 default)")
    ap.add_argument("--xcd", type=int, default=1, help="xcd_remap option")
    ap.add_argument("--batch", type=int, default=0, help="quartets per scan/SVD batch (0 = library default 2^20)")
    ap.add_argument("--sort", default="none", choices=["none", "ab", "lex", "same"], help="order of the quartet list")
    args = ap.parse_args()
    import torch
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine

    T, S, _ = synth.CONFIGS[args.config]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[args.config])
    Q = args.quartets
    quartets = synth.random_quartets(T, Q, 4242)
    if args.sort == "ab":
        quartets = quartets[np.argsort(quartets[:, 0].astype(np.int64) * T + quartets[:, 1], kind="stable")]
    elif args.sort == "same":      # diagnostic: every quartet reads the same four rows (cache-hot)
        quartets = np.tile(np.array([[3, 17, 40, 99]], np.uint32), (Q, 1))
    elif args.sort == "lex":
        quartets = quartets[np.lexsort((quartets[:, 3], quartets[:, 2], quartets[:, 1], quartets[:, 0]))]
    dev = torch.device("cuda:0")
    eng = QuartetEngine(0)
    eng.set_data(tmparr, tmpmap)
    eng.set_option("xcd_remap", args.xcd)
    eng.set_option("batch", args.batch)
    d_q = torch.from_numpy(quartets.astype(np.int32)).to(dev)
    d_rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
    d_rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    variants = list(itertools.product(
        args.modes.split(","), [int(x) for x in args.nreps.split(",")],
        [int(x) for x in args.wpcs.split(",")], [int(x) for x in args.methods.split(",")],
        [int(x) for x in args.orders.split(",")], [int(x) for x in args.svds.split(",")],
        [int(x) for x in args.wgs.split(",")], [int(x) for x in args.svdwpcs.split(",")]))
    times = {v: [] for v in variants}
    eng.timing_enable(True)
    for rnd in range(args.rounds + 1):
        for v in variants:
            mode, nrep, wpc, meth, order, svdm, wgm, swpc = v
            eng.set_option("svd_wpc", swpc)
            eng.set_option("scan_wg", wgm)
            eng.set_option("svd_method", svdm)
            eng.set_option("order", order)
            eng.set_option("scan_method", meth)
            eng.set_option("nrep", nrep)
            eng.set_option("waves_per_cu", wpc)
            eng.resolve_dev(d_q.data_ptr(), Q, mode == "sub", d_rstat.data_ptr(), d_rscor.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            tot, scan, svd, n = eng.timing_read_split()
            if rnd:                     # round 0 is warm-up
                times[v].append((tot, scan, svd))
    print(f"# {args.config} T={T} S={S} Q={Q} rounds={args.rounds} sort={args.sort}")
    print("mode nrep wpc meth ord svd  wg swpc total_ms  scan_ms   svd_ms   Mq/s")
    for v in variants:
        t = np.array(times[v]).min(axis=0)
        print(f"{v[0]:4s} {v[1]:4d} {v[2]:3d} {v[3]:4d} {v[4]:3d} {v[5]:3d} {v[6]:3d} {v[7]:4d} {t[0]:9.3f} {t[1]:8.3f} {t[2]:8.3f} {Q / t[0] / 1e3:7.2f}")


if __name__ == "__main__":
    main()
