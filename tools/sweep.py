#!/usr/bin/env python
"""Interleaved A/B timing of engine options in ONE process.

usage: python tools/sweep.py [--config c3] [--quartets 1000000] [--rounds 3] [--mode sub|full]
                             [--opt name=v1,v2 ...] [--api dev|host]
Every --opt adds an axis; the cartesian product is timed round-robin (round 0 = warm-up).  Prints one line
per variant: min wall ms of a whole pass (device-resident `tq_resolve_dev`, or `tq_resolve_to_host` with
--api host), and the per-kernel HIP-event milliseconds of that pass (tq_timing_read_kernels).
"""
import argparse
import itertools
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--quartets", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--mode", default="sub")
    ap.add_argument("--api", default="dev", choices=["dev", "host"])
    ap.add_argument("--sort", default="none", choices=["none", "lex"], help="order of the quartet list")
    ap.add_argument("--opt", action="append", default=[], help="name=v1,v2,...  (engine option axis)")
    ap.add_argument("--lib", default="", help="load this build of the library instead of the in-tree one (A/B of two builds)")
    args = ap.parse_args()
    import torch
    from tetrad_amd import synth
    if args.lib:
        from tetrad_amd import _lib
        _lib.LIB_PATH = Path(args.lib).resolve()
    from tetrad_amd.engine import QuartetEngine, pinned_empty

    T, S, _ = synth.CONFIGS[args.config]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[args.config])
    Q = args.quartets
    dev = torch.device("cuda:0")
    eng = QuartetEngine(0)
    eng.set_data(tmparr, tmpmap)
    stream = torch.cuda.current_stream().cuda_stream
    idx = np.random.default_rng(4242).choice(synth.comb(T, 4), size=Q, replace=False).astype(np.int64)
    if args.sort == "lex":
        idx.sort()
    d_r = torch.from_numpy(idx).to(dev)
    d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
    eng.unrank_dev(d_r.data_ptr(), Q, d_q.data_ptr(), stream)
    d_rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
    d_rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
    out = (pinned_empty((Q, 2), np.uint32), pinned_empty((Q, 3), np.float64), pinned_empty(Q, np.uint8))
    axes = []
    for o in args.opt:
        name, vals = o.split("=")
        axes.append([(name, int(v)) for v in vals.split(",")])
    variants = list(itertools.product(*axes)) if axes else [()]
    sub = args.mode == "sub"
    res = {v: [] for v in variants}
    eng.timing_enable(True)
    for rnd in range(args.rounds + 1):
        for v in variants:
            for name, val in v:
                eng.set_option(name, val)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if args.api == "dev":
                eng.resolve_dev(d_q.data_ptr(), Q, sub, d_rstat.data_ptr(), d_rscor.data_ptr(), 0, stream)
                torch.cuda.synchronize()
            else:
                eng.resolve_to_host(d_q.data_ptr(), Q, sub, out=out)
            wall = (time.perf_counter() - t0) * 1e3
            k, _ = eng.timing_read_kernels()
            if rnd:
                res[v].append([wall] + [k[t] for t in eng.KERNEL_TAGS])
    print(f"# {args.config} T={T} S={S} Q={Q} mode={args.mode} api={args.api} sort={args.sort} rounds={args.rounds}")
    print(f"{'variant':48s}  wall_ms   order    scan  bidiag   bdsqr   score   Mq/s")
    for v in variants:
        t = np.array(res[v]).min(axis=0)
        name = " ".join(f"{n}={x}" for n, x in v) or "default"
        print(f"{name:48s} {t[0]:8.3f} {t[1]:7.3f} {t[2]:7.3f} {t[3]:7.3f} {t[4]:7.3f} {t[5]:7.3f} {Q / t[0] / 1e3:6.1f}")


if __name__ == "__main__":
    main()
