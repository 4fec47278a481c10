"""Randomised GPU-vs-oracle comparison over shapes, sparsity, locus structure, batch sizes and options.
Usage: python tests/fuzz_gpu.py [seconds] [seed]   (prints one line per case, exits 1 on the first mismatch)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
eng = QuartetEngine(0)
t_end = time.time() + budget
case = 0
while time.time() < t_end:
    case += 1
    T = int(rng.integers(4, 48)) if rng.random() < 0.9 else int(rng.integers(48, 400))
    S = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 4097, int(rng.integers(1, 7000))]))
    p = float(rng.choice([0.005, 0.02, 0.05, 0.2]))
    missing = float(rng.choice([0.0, 0.05, 0.3, 0.7, 0.95]))
    tmparr, tmpmap = synth.simulate_tmparr(T, S, int(rng.integers(1 << 30)), p=p, missing=missing)
    style = int(rng.integers(4))
    if style == 1:                       # one locus per site
        tmpmap[:, 0] = np.arange(S)
    elif style == 2:                     # a single locus
        tmpmap[:, 0] = 0
    elif style == 3:                     # long runs with arbitrary (non-consecutive, non-monotone) ids
        runs = np.cumsum(rng.integers(1, 200, size=S))
        ids = rng.permutation(S + 5)[:S].astype(np.uint32)
        tmpmap[:, 0] = ids[np.searchsorted(runs, np.arange(S), side="right")]
    if rng.random() < 0.3:
        tmparr[rng.integers(T)] = 78     # an all-missing taxon
    Q = int(rng.choice([1, 3, 63, 64, 65, 255, 1000, 1024, 1025, int(rng.integers(1, 6000))]))
    if rng.random() < 0.03:
        Q = int(rng.integers(32768, 36000))      # large enough for the device sort + the shared-row image in sorted order
    q = np.sort(np.stack([rng.permutation(T)[:4] for _ in range(Q)]), axis=1).astype(np.uint32)
    if rng.random() < 0.5:
        q = q[np.lexsort((q[:, 3], q[:, 2], q[:, 1], q[:, 0]))]
    opts = {"scan_wg": int(rng.choice([0, 0, 1, 2, 8])), "batch": int(rng.choice([0, 0, 700])),
            "svd_method": int(rng.choice([1, 1, 0])), "xcd_remap": int(rng.integers(2)),
            "svd_chunk": int(rng.choice([0, 0, 1, 100, 1024])), "svd_streams": int(rng.choice([0, 0, 1])),
            "share_c": int(rng.choice([0, 0, 1])), "park_t": int(rng.choice([1, 1, 0])), "scan_pair": int(rng.choice([0, 0, 1])),
            "scan_method": int(rng.choice([-1, -1, 0, 1, 6])), "bidiag_layout": int(rng.choice([1, 1, 0])), "wg_min_quartets": int(rng.choice([0, 64, 64])),
            "scan_dp": int(rng.choice([1, 1, 0])), "scan_f4": int(rng.choice([-1, -1, 0, 1])), "dp_min_quartets": int(rng.choice([0, 2, 2, 500]))}
    if rng.random() < 0.35:              # the option set under which full-mode batches go to the joint-histogram scan (scan_dp.hpp)
        opts.update(scan_wg=0, scan_method=-1, share_c=0, scan_pair=0, scan_dp=1, scan_f4=-1, dp_min_quartets=2)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.set_data(tmparr, tmpmap)
    if rng.random() < 0.25 and T >= 4:
        # lexicographic rank range resolved on the device == the same quartets through the host API
        import torch
        from math import comb
        total = comb(T, 4)
        n = int(min(total, rng.integers(1, 3000)))
        first = int(rng.integers(0, total - n + 1))
        dev = torch.device("cuda:0")
        dq = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        drs = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        dsc = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        dfl = torch.zeros(n, dtype=torch.uint8, device=dev)
        sub0 = bool(rng.integers(2))
        eng.resolve_range_dev(first, n, sub0, dq.data_ptr(), drs.data_ptr(), dsc.data_ptr(), dfl.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        qq = dq.cpu().numpy().astype(np.uint32)
        want_q = synth.unrank_quartets(np.arange(first, first + n, dtype=np.uint64), T)
        r2 = eng.resolve(qq, sub0)
        same = (np.array_equal(qq, want_q) and np.array_equal(drs.cpu().numpy().astype(np.uint32), r2[0])
                and np.array_equal(dsc.cpu().numpy(), r2[1]) and np.array_equal(dfl.cpu().numpy(), r2[2]))
        print(f"case {case} range first={first} n={n} sub={sub0} {'ok' if same else 'MISMATCH'}", flush=True)
        if not same:
            sys.exit(1)
    for sub in (True, False):
        rstat, rscor, flags = eng.resolve(q, sub)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
        smax = np.maximum(o["svds"].max(axis=(1, 2)), 1e-300)[:, None]
        ok_n = np.array_equal(rstat[:, 1], o_rstat[:, 1])
        zero = o_rstat[:, 1] == 0
        err = np.abs(rscor - o_rscor) / (np.abs(o_rscor) + 1e-6 * smax)
        ok_s = bool((err[~zero] < 1e-6).all()) if (~zero).any() else True
        plain = ((flags | o["flags"]) & 3) == 0
        ok_t = np.array_equal(rstat[plain, 0], o_rstat[plain, 0])
        ok_f = np.array_equal(flags & 1, o["flags"] & 1)
        status = "ok" if (ok_n and ok_s and ok_t and ok_f) else "MISMATCH"
        print(f"case {case} T={T} S={S} p={p} miss={missing} loc={style} Q={Q} sub={sub} {opts} "
              f"flagged={int((~plain).sum())} maxerr={err[~zero].max() if (~zero).any() else 0:.1e} {status}", flush=True)
        if status != "ok":
            np.savez("/tmp/fuzz_fail.npz", tmparr=tmparr, tmpmap=tmpmap, q=q, sub=sub)
            print(ok_n, ok_s, ok_t, ok_f)
            sys.exit(1)
print(f"{case} cases clean")
