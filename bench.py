#!/usr/bin/env python
"""bench.py -- resolved quartets/s of the per-quartet hot path on N MI355X.

One "step" = one pass of the hot path (tq_resolve_dev: site scan -> 16x16 count matrices ->
3 singular-value decompositions -> scores/topology) over one batch of synthetic quartets that
is already resident in HBM, plus -- for N > 1 -- the RCCL all-gather of the result records.

Workload at N=1: BASELINE.json configs[2] ("c3"): 128 taxa, 50k SNPs, 1e6 random quartets,
subsample_snps=True (the reference's default, resolve_quartets.py:21).  Weak scaling: every
rank resolves its own contiguous 1e6-quartet shard of an N x 1e6 global sample.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HIP-event
kernel time on the launch stream vs algorithmic bytes, SURVEY.md 8d: A = 4*S + 48 B/quartet)
and `cpu_baseline` (the oracle -- a port of the reference worker -- timed on host cores).

The hot path is a short kernel chain per batch: ordering (radix sort by the first two taxa) +
tq_scan_wg_kernel (site scan -> 256 pattern counts per quartet), then the singular-value stage
tq_bidiag_kernel + tq_bdsqr_kernel + tq_score_kernel (or tq_svd_kernel, the Jacobi path).
`roofline.achieved` prices the algorithmic bytes against the SUM of all of them (HIP events on
the launch stream bracket the scan stage and the singular-value stage of every pass); the two
stage durations are listed beside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def _cpu_worker(args):
    """cpu_baseline leg: the oracle (port of resolve_quartets.py:191-265) on one host core."""
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import oracle as orc
    tmparr, tmpmap, quartets, sub = args[:4]
    tuned = len(args) > 4 and args[4]
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(1)
    except Exception:  # pragma: no cover
        ctx = None
    t0 = time.perf_counter()
    fn = orc.new_infer_resolved_quartets_batched if tuned else orc.new_infer_resolved_quartets
    _, rstat, rscor = fn(tmparr, tmpmap, quartets, sub)
    dt = time.perf_counter() - t0
    return dt, rstat, rscor


def cpu_baseline(tmparr, tmpmap, quartets, sub, gpu_rstat, gpu_rscor, budget_s=15.0):
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    # calibrate on one core, then size the sample for ~budget_s of wall time on all cores
    dt, _, _ = _cpu_worker((tmparr, tmpmap, quartets[:300], sub))
    per_core = 300 / dt
    n = int(min(len(quartets), max(cores * 200, per_core * cores * budget_s)))
    chunks = np.array_split(np.arange(n), cores)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(tmparr, tmpmap, quartets[c], sub) for c in chunks])
    # the slowest worker's own clock: pool start-up (forking a process that holds GPU mappings) is not CPU work
    wall = min(time.perf_counter() - t0, max(r[0] for r in res))
    rstat = np.concatenate([r[1] for r in res])
    rscor = np.concatenate([r[2] for r in res])
    # the oracle doubles as the checker on this sample
    parity = dict(
        n=n,
        nsnps_equal=bool(np.array_equal(rstat[:, 1], gpu_rstat[:n, 1])),
        topology_equal=bool(np.array_equal(rstat[:, 0], gpu_rstat[:n, 0])),
        score_max_rel_err=float(np.max(np.abs(rscor - gpu_rscor[:n]) / np.abs(rscor))),
    )
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # second figure: the same path as a tuned CPU implementation would run it -- one
    # values-only LAPACK SVD per matrix for a whole chunk instead of the reference's two full ones
    n2 = max(cores * 100, n // 4)
    chunks2 = np.array_split(np.arange(n2), cores)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res2 = pool.map(_cpu_worker, [(tmparr, tmpmap, quartets[c], sub, True) for c in chunks2])
    wall2 = min(time.perf_counter() - t0, max(r[0] for r in res2))
    rstat2 = np.concatenate([r[1] for r in res2])
    parity["tuned_cpu_variant_topology_equal"] = bool(np.array_equal(rstat2[:, 0], gpu_rstat[:n2, 0]))
    return dict(value=n / wall, unit="quartets/s", cores=cores, cpu_model=model, kind="port",
                tuned_variant=dict(value=n2 / wall2, unit="quartets/s", cores=cores,
                                   what=f"first {n2} quartets, compiled count loop + one batched values-only "
                                        f"numpy.linalg.svd per chunk (no interpreter in the per-quartet loop "
                                        f"beyond one ctypes call)"),
                sample=f"first {n} quartets of the same batch, {cores} processes x 1 thread, "
                       f"oracle.new_infer_resolved_quartets (C count loop + numpy.linalg svd/matrix_rank)",
                per_core=per_core), parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4"])
    ap.add_argument("--quartets", type=int, default=0, help="quartets per GPU (0 = config default)")
    ap.add_argument("--full", action="store_true", help="subsample_snps=False")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 control flow on a box with fewer GPUs than ranks")
    ap.add_argument("--nrep", type=int, default=0)
    ap.add_argument("--order", type=int, default=-1, help="0 = natural quartet order, 1 = (a,b)-sorted (default)")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--phases", type=int, default=0, help="diagnostic: 1 scan only, 2 SVD only (invalid as a result)")
    args = ap.parse_args()

    import torch
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    T, S, qdef = synth.CONFIGS[args.config]
    sub = not args.full
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[args.config])
    Q = args.quartets or qdef or int(synth.comb(T, 4))
    if qdef is None and not args.quartets:
        quartets = synth.all_quartets(T)
    else:
        # one global sample of world*Q quartets, rank r owns the contiguous slice [r*Q, (r+1)*Q)
        idx = np.random.default_rng(synth.CONFIG_SEEDS[args.config] + 1000).choice(
            synth.comb(T, 4), size=world * Q, replace=False)
        quartets = synth.unrank_quartets(idx[rank * Q:(rank + 1) * Q], T)

    eng = QuartetEngine(dev_index)
    if args.nrep:
        eng.set_option("nrep", args.nrep)
    if args.waves_per_cu:
        eng.set_option("waves_per_cu", args.waves_per_cu)
    if args.phases:
        eng.set_option("phases", args.phases)
    if args.order >= 0:
        eng.set_option("order", args.order)
    eng.set_data(tmparr, tmpmap)

    d_q = torch.from_numpy(quartets.astype(np.int32)).to(dev)
    # two sets of output buffers: for N > 1 the all-gather of pass i runs (on RCCL's stream) while
    # the kernels of pass i+1 write the other set
    nbuf = 2 if world > 1 else 1
    # one contiguous 32-byte-per-quartet slab per buffer set: [rstat u32[Q,2] | rscor f64[Q,3]], so that the
    # result gather is ONE collective per step with no packing kernel (fixed-size records)
    d_outs = [torch.zeros(32 * Q, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    d_rstats = [o[:8 * Q].view(torch.int32).view(Q, 2) for o in d_outs]
    d_rscors = [o[8 * Q:].view(torch.float64).view(Q, 3) for o in d_outs]
    d_flags = torch.zeros(Q, dtype=torch.uint8, device=dev)
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        g_outs = [torch.zeros(world * 32 * Q, dtype=torch.uint8, device=gdev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream().cuda_stream
    pending = [[], []]
    counter = [0]

    def step():
        b = counter[0] % nbuf
        counter[0] += 1
        for w in pending[b]:           # the gather that last read this buffer set must be done
            w.wait()
        pending[b] = []
        eng.resolve_dev(d_q.data_ptr(), Q, sub, d_rstats[b].data_ptr(), d_rscors[b].data_ptr(),
                        d_flags.data_ptr(), stream)
        if world > 1:
            # result gather: every rank ends up with the whole node's rows in global order
            pending[b] = [dist.all_gather_into_tensor(g_outs[b], d_outs[b].to(gdev), async_op=True)]

    def fence():
        for ws in pending:
            for w in ws:
                w.wait()
            ws.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, scan_ms, svd_ms, launches = eng.timing_read_split()
    eng.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # PCIe-inclusive rate of the host-buffer API (quartets H2D + results D2H per call); reported
    # beside `value`, never as `value`
    eng.resolve(quartets, sub)          # first call allocates the host-API scratch
    t1 = time.perf_counter()
    eng.resolve(quartets, sub)
    pcie_inclusive = Q / (time.perf_counter() - t1)

    # device-to-device copy rate of this GPU (what "8 TB/s" means in practice here; SURVEY 8d asks
    # for the vendor figure and a measured one side by side): 1 GiB read + 1 GiB written per copy
    hbm_copy = None
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        hbm_copy = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) / 1e3) / 1e9
        del src, dst

    last = (counter[0] - 1) % nbuf
    rstat = d_rstats[last].cpu().numpy().astype(np.uint32)
    rscor = d_rscors[last].cpu().numpy()
    flags = d_flags.cpu().numpy()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * Q * args.steps / elapsed
        A = 4 * S + 48                                   # algorithmic bytes per quartet
        bytes_per_launch = Q * A + 4 * S
        avg_kernel_s = kernel_ms / max(1, launches) / 1e3
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        traffic = None
        tf = REPO / "profiles" / f"traffic_{args.config}_{'sub' if sub else 'full'}.json"
        if tf.exists():
            traffic = json.loads(tf.read_text()).get("hbm_bytes_per_launch")
        line = {
            "metric": "resolved quartets/sec (whole node)", "value": value, "unit": "quartets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 scan / u32 counts / f64 SVD", "data": "synthetic",
            "config": {"workload": f"{args.config}: {T} taxa x {S} SNPs, {Q} {'random' if (qdef or args.quartets) else 'lexicographic (all)'} quartets per GPU, "
                                   f"subsample_snps={sub}", "quartets_per_gpu": Q, "taxa": T, "snps": S,
                       "subsample_snps": sub, "parallelism": f"quartet-sharded x{world} + all-gather"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "hbm_copy_measured_GBs": hbm_copy,
                         "kernel": "one pass of the hot path: tq_scan_wg_kernel (+ordering) then tq_bidiag/tq_bdsqr/tq_score",
                         "kernel_ms": kernel_ms / max(1, launches),
                         "scan_stage_ms": scan_ms / max(1, launches),
                         "svd_stage_ms": svd_ms / max(1, launches),
                         "achieved_scan_stage_only": bytes_per_launch / (scan_ms / max(1, launches) / 1e3) / 1e9,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "pcie_inclusive_value_per_gpu": pcie_inclusive,
            "flags": {"zero_data": int((flags & 1).sum()), "degenerate": int(((flags & 2) > 0).sum())},
        }
        if args.phases in (1, 2):
            line["INVALID_diagnostic_phases"] = args.phases
        if not args.no_cpu and args.phases in (0, 3) and world == 1:     # CPU leg: rank 0 at N=1 only
            cb, parity = cpu_baseline(tmparr, tmpmap, quartets, sub, rstat, rscor)
            line["cpu_baseline"] = cb
            line["parity_on_cpu_sample"] = parity
            line["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
