#!/usr/bin/env python
"""bench.py -- resolved quartets/s of the per-quartet hot path on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic quartets: ordering + site scan
-> 16x16 count matrices -> 3 singular-value decompositions -> scores/topology, WITH the results
delivered to host arrays inside the step ("wall time of tq_resolve over the full batch
incl. result D2H and, for N>1, the all-gather").  Inputs (genotype matrix, quartets) are resident in
HBM when the timed region starts.  Every step is synchronous: when it ends, the rows are on the host
(on rank 0 for N > 1).

Workloads (BASELINE.json configs):
  N = 1  (default)      c3: 128 taxa x 50k SNPs, 1e6 random quartets, subsample_snps=True, through
                        tq_resolve_to_host.  `--config c2|c4` select the other single-GPU shapes.
  N > 1  (default)      c4: 256 taxa x 100k SNPs, ONE seeded sample of 5e6 quartets cut over the N ranks
                        (strong scaling), quartet parts scanned per rank, results all-gathered (RCCL) piece by
                        piece, regrouped into global order and copied to rank 0's host arrays.
         --weak         every rank owns 1e6 quartets of an N x 1e6 sample of the c3 matrix.
  --config c5           c3 shape, K bootstrap replicates: per step one replicate built on the device
                        (locus resample + within-locus shuffle + IUPAC resolution), a fresh 1e6-quartet
                        sample, resolve (sharded over N), gather, D2H.  RNG draws in the reference's
                        order on one Generator (run_inference.py:378-407), made ahead by a producer thread.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (algorithmic bytes per
launch over the HIP-event kernel time on the launch stream, SURVEY.md 8d: A = 4*S + 48 B/quartet, plus a
per-kernel list with the fraction of the roof that actually binds each kernel) and `cpu_baseline` (the
oracle -- a port of the reference worker -- timed on host cores; N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (tools/probe_valu.hip measures 74)
L2_TO_CU_PEAK_GBS = 17800.0    # MI355X_MICROARCH.md "Indexed rows": rows shared out of the XCD's L2, 16.8-18.8 TB/s


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
    n = int(min(len(quartets), max(cores * 200, 300 / dt * cores * budget_s)))
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
                per_core=n / wall / cores), parity


def git_head() -> str:
    try:
        return subprocess.check_output(["git", "-C", str(REPO), "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        f = REPO / ".build_commit"          # written by __graft_entry__.build() where git is available
        return f.read_text().strip() if f.exists() else "unknown"


def kernel_rooflines(kms: dict, calls: int, Q: int, S: int, sub: bool, cfg: str):
    """roofline.kernels[]: per kernel the measured ms per pass (HIP events of THIS run) and the fraction
    (<= 1) of the roof that binds it, from a stated model of the work it has to do:
      scan    bytes it has to pull through the L2 -> CU path (4.4 KiB per quartet and 2048-site step: 3.5 KiB
              of its own rows c,d + a quarter of the 3.75 KiB shared (a,b) image) against the
              fabric rate for rows served from the XCD's L2 (MI355X_MICROARCH.md);
      bidiag  useful f64 flops of a Householder bidiagonalisation, (8/3) n^3 = 10 923 per 16x16 matrix;
      bdsqr   useful f64 flops of the implicit-shift QR sweeps: rotation steps per matrix (measured with a
              counting build on this workload, profiles/) x 26 flop per step
              (two Givens generations + their applications);
      against the f64 vector peak.  The busy-cycle counters (VALU / LDS) of the same kernels are in the
      committed rocprofv3 --pmc summaries; `pmc` quotes them with their source file when present."""
    calls = max(1, calls)
    per = {k: v / calls for k, v in kms.items()}
    out = []
    steps = -(-S // 2048)
    scan_bytes = Q * steps * (3.5 + 3.75 / 4) * 1024
    if per.get("order", 0) > 0:
        out.append(dict(name="ordering (tq_key_kernel + radix sort)", ms=per["order"], bound="latency (~20 small launches)",
                        frac=None))
    if per.get("scan", 0) > 0:
        ach = scan_bytes / (per["scan"] / 1e3) / 1e9
        out.append(dict(name="tq_scan_wg_kernel", ms=per["scan"],
                        bound="co-bound: l2->cu fabric (this model), LDS pipe and VALU issue (pmc shares below); cutting the "
                              "bytes by 40 % made it slower (DESIGN.md 4.1)",
                        achieved=ach, peak=L2_TO_CU_PEAK_GBS, unit="GB/s", frac=ach / L2_TO_CU_PEAK_GBS,
                        model="4.4 KiB per quartet-step through the L2->CU path"))
    if per.get("bidiag", 0) > 0:
        fl = 3 * Q * (8.0 / 3.0) * 16 ** 3
        ach = fl / (per["bidiag"] / 1e3) / 1e12
        out.append(dict(name="tq_bidiag_kernel", ms=per["bidiag"], bound="f64 valu", achieved=ach,
                        peak=F64_VECTOR_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / F64_VECTOR_PEAK_TFLOPS,
                        model="(8/3)*16^3 useful flop per matrix"))
    if per.get("bdsqr", 0) > 0:
        # measured with the kernel's counting mode (tools/bdsqr_stats.py, profiles/r02_pipeline/bdsqr_work.txt)
        steps_per_matrix = {"c2": 235.4, "c3": 226.6, "c4": 221.1}.get(cfg, 226.6) if sub else \
            {"c2": 229.0, "c3": 220.6, "c4": 215.6}.get(cfg, 220.6)
        fl = 3 * Q * steps_per_matrix * 26
        ach = fl / (per["bdsqr"] / 1e3) / 1e12
        out.append(dict(name="tq_bdsqr_kernel", ms=per["bdsqr"], bound="f64 valu", achieved=ach,
                        peak=F64_VECTOR_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / F64_VECTOR_PEAK_TFLOPS,
                        model=f"{steps_per_matrix:.1f} rotation steps per matrix x 26 useful flop (a wave issues 1.33x that: "
                              f"in every sweep it runs for its longest block)"))
    if per.get("score", 0) > 0:
        by = Q * (3 * 16 * 8 + 4 + 16 + 33)
        ach = by / (per["score"] / 1e3) / 1e9
        out.append(dict(name="tq_score_kernel", ms=per["score"], bound="hbm", achieved=ach, peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=ach / HBM_PEAK_GBS, model="437 B per quartet read + written"))
    pmc = REPO / "profiles" / "pmc_busy_latest.json"
    if pmc.exists():
        try:
            info = json.loads(pmc.read_text())
            for k in out:
                for name, v in info.get("kernels", {}).items():
                    if name in k["name"]:
                        k["pmc"] = dict({x: v[x] for x in ("valu_busy", "lds_busy", "lds_conflict_share") if x in v},
                                        source="profiles/pmc_busy_latest.json: " + str(info.get("source")),
                                        commit=info.get("commit"))
        except Exception:
            pass
    return out


def hbm_copy_rate(torch, dev):
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) / 1e3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=["c2", "c3", "c4", "c5"],
                    help="default: c3 on one GPU, c4 (strong scaling) on several")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank its own 1e6 quartets of the c3 matrix")
    ap.add_argument("--quartets", type=int, default=0, help="quartets in the batch (0 = config default)")
    ap.add_argument("--full", action="store_true", help="subsample_snps=False")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 control flow on a box with fewer GPUs than ranks")
    ap.add_argument("--pieces", type=int, default=0, help="N > 1: result pieces per batch (0 = automatic)")
    ap.add_argument("--sharded", action="store_true",
                    help="take the N > 1 code path (process group, all-gather per piece, regrouping, verification) "
                         "even with one rank: a one-GPU rehearsal of the RCCL path when launched through torch.distributed.run")
    ap.add_argument("--sampler", default="host", choices=["host", "device"],
                    help="c5: quartet sample drawn on the project Generator (reference stream) or on the device")
    ap.add_argument("--svd-chunk", type=int, default=0)
    ap.add_argument("--order", type=int, default=-1, help="0 = natural quartet order, 1 = sorted (default)")
    ap.add_argument("--phases", type=int, default=0, help="diagnostic: 1 scan only, 2 SVD only (invalid as a result)")
    args = ap.parse_args()

    import torch
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine, pinned_empty

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
    multi = world > 1 or (args.sharded and "RANK" in os.environ)
    if multi:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg = args.config or ("c3" if not multi else ("c3" if args.weak else "c4"))
    if args.weak and cfg != "c3":
        raise SystemExit("--weak is the c3 shape")
    sub = not args.full
    eng = QuartetEngine(dev_index)
    if args.phases:
        eng.set_option("phases", args.phases)
    if args.order >= 0:
        eng.set_option("order", args.order)
    if args.svd_chunk:
        eng.set_option("svd_chunk", args.svd_chunk)

    if cfg == "c5":
        return bench_c5(args, eng, torch, dist, dev, world, rank)

    T, S, qdef = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    eng.set_data(tmparr, tmpmap)
    lexi = qdef is None and not args.quartets
    if lexi:
        Q = int(synth.comb(T, 4))
        ranks_all = np.arange(Q, dtype=np.int64)
    else:
        Q = args.quartets or qdef
        if args.weak:
            Q *= world
        # ONE seeded sample for the whole job, identical on every rank (combinations.py:109-114)
        ranks_all = np.random.default_rng(synth.CONFIG_SEEDS[cfg] + 1000).choice(
            synth.comb(T, 4), size=Q, replace=False).astype(np.int64)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    if not multi:
        d_r = torch.from_numpy(ranks_all).to(dev)
        d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
        eng.unrank_dev(d_r.data_ptr(), Q, d_q.data_ptr(), stream)
        torch.cuda.synchronize()
        out = (pinned_empty((Q, 2), np.uint32), pinned_empty((Q, 3), np.float64), pinned_empty(Q, np.uint8))

        def step():
            eng.resolve_to_host(d_q.data_ptr(), Q, sub, out=out)

        def results():
            return out
    else:
        from tetrad_amd.distributor import ShardedResolver
        res = ShardedResolver(Q, engine=eng, device=dev_index, dst=0, pieces=args.pieces or None,
                              collective_always=True)
        res.set_ranks(ranks_all)
        torch.cuda.synchronize()
        last = [None]

        def step():
            last[0] = None                  # hand the previous rows back to the pinned pool before new ones are taken
            res.start(sub)
            last[0] = res.finish()

        def results():
            return last[0]

    for _ in range(args.warmup):
        step()
    barrier()
    eng.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kms, launches = eng.timing_read_kernels()
    # per-kernel durations of the singular-value stage: in the timed region two chunks run concurrently on
    # two streams (one chunk's tail is filled by the next), so their event spans overlap; three more steps
    # with the chunks serialised on one stream give each kernel's own duration for roofline.kernels[]
    eng.set_option("svd_streams", 1)
    step()
    eng.timing_read_kernels()
    for _ in range(3):
        step()
    kms_serial, launches_serial = eng.timing_read_kernels()
    eng.set_option("svd_streams", 0)
    eng.timing_enable(False)
    if multi:
        gdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step()                                  # results of the default configuration

    rstat, rscor, flags = results()
    if not multi:
        # device-resident rate (no result D2H) and the PCIe-inclusive rate of the host-buffer API (quartets
        # H2D from a page-locked array + results D2H per call); reported beside `value`, never as `value`
        quartets_h = pinned_empty((Q, 4), np.uint32)
        quartets_h[...] = d_q.cpu().numpy().view(np.uint32)
        d_rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
        d_rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
        d_flags = torch.zeros(Q, dtype=torch.uint8, device=dev)
        eng.resolve_dev(d_q.data_ptr(), Q, sub, d_rstat.data_ptr(), d_rscor.data_ptr(), d_flags.data_ptr(), stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            eng.resolve_dev(d_q.data_ptr(), Q, sub, d_rstat.data_ptr(), d_rscor.data_ptr(), d_flags.data_ptr(), stream)
        torch.cuda.synchronize()
        extra["device_resident_value"] = 3 * Q / (time.perf_counter() - t1)
        r_h = eng.resolve(quartets_h, sub)     # two warm calls: the pool then holds both sets of result arrays that
        r_h = eng.resolve(quartets_h, sub)     # are alive at a time (pinning fresh pages costs milliseconds)
        t1 = time.perf_counter()
        for _ in range(3):
            r_h = eng.resolve(quartets_h, sub)
        extra["pcie_inclusive_value"] = 3 * Q / (time.perf_counter() - t1)
        extra["host_api_equals_timed_path"] = bool(np.array_equal(r_h[0], rstat) and np.array_equal(r_h[1], rscor))
        q_pageable = np.array(quartets_h)
        out_pg = (np.empty((Q, 2), np.uint32), np.empty((Q, 3), np.float64), np.empty(Q, np.uint8))
        import ctypes
        def pageable_call():
            rc = eng._lib.tq_resolve(eng._h, ctypes.c_void_p(q_pageable.ctypes.data), Q, int(sub),
                                     ctypes.c_void_p(out_pg[0].ctypes.data), ctypes.c_void_p(out_pg[1].ctypes.data),
                                     ctypes.c_void_p(out_pg[2].ctypes.data))
            assert rc == 0
        pageable_call()
        t1 = time.perf_counter()
        for _ in range(3):
            pageable_call()
        extra["pcie_inclusive_value_pageable_arrays"] = 3 * Q / (time.perf_counter() - t1)
    else:
        # the gathered rows on rank 0 against what every rank computed itself: a checksum over each rank's
        # own rows and 64 sampled rows per rank, compared bit for bit
        P = res.plan
        own = res.device_rows()
        idx = torch.from_numpy(P.local_index(rank)).to(own[0].device)
        mine = [x.index_select(0, idx) for x in own] if len(idx) else None
        pick = np.random.default_rng(rank).integers(0, max(1, len(idx)), size=min(64, len(idx)))
        rec = dict(rank=rank, n=int(len(idx)),
                   nsnps_sum=int(mine[0][:, 1].to(torch.int64).sum().item()) if mine else 0,
                   score_bits_xor=int(np.bitwise_xor.reduce(mine[1].cpu().numpy().view(np.int64).ravel())) if mine else 0,
                   rows=[(int(P.local_index(rank)[i]), mine[0][i].tolist(), mine[1][i].cpu().numpy().view(np.int64).tolist())
                         for i in pick] if mine else [])
        recs = [None] * world
        dist.all_gather_object(recs, rec)
        if rank == 0:
            ok = True
            for r in recs:
                gi = P.local_index(r["rank"])
                ok &= int(rstat[gi, 1].astype(np.int64).sum()) == r["nsnps_sum"]
                ok &= int(np.bitwise_xor.reduce(rscor[gi].view(np.int64).ravel())) == r["score_bits_xor"] if len(gi) else True
                for g, rs, sc in r["rows"]:
                    ok &= rstat[g].astype(np.int64).tolist() == [x & 0xFFFFFFFF for x in rs]
                    ok &= rscor[g].view(np.int64).tolist() == sc
            extra["gather_verified"] = bool(ok)
            extra["gather_pieces"] = P.npieces
            extra["rows_per_piece_per_rank"] = P.part
            # the SAME workload on one GPU, for a like-for-like scaling figure (the default N = 1 run is c3, a
            # lighter shape): quoted from the committed one-GPU line of this config, not measured in this run
            ref = REPO / "profiles" / "r02_pipeline" / f"bench_{cfg}_n1.json"
            if ref.exists() and not args.quartets and not args.weak:
                try:
                    r1 = json.loads(ref.read_text())
                    extra["same_workload_on_one_gpu"] = dict(value=r1["value"], commit=r1.get("commit"),
                                                             source=f"profiles/r02_pipeline/{ref.name} (not measured in this run)")
                except Exception:
                    pass

    hbm_copy = hbm_copy_rate(torch, dev) if rank == 0 else None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = Q * args.steps / elapsed
        A = 4 * S + 48                                   # algorithmic bytes per quartet
        q_rank = res.n_local if multi else Q               # quartets one launch of the scan kernel covers here
        bytes_per_launch = q_rank * A + 4 * S
        per_pass = {k: v / max(1, launches) for k, v in kms.items()}
        dominant_ms = per_pass["scan"]
        achieved = bytes_per_launch / (dominant_ms / 1e3) / 1e9
        traffic, traffic_src = None, "not measured in this run"
        tf = REPO / "profiles" / f"traffic_{cfg}_{'sub' if sub else 'full'}.json"
        if tf.exists() and not multi and not args.quartets:
            tj = json.loads(tf.read_text())
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, collated in {tf.name} "
                           f"(commit {tj.get('commit', 'unknown')}); not re-measured in this run")
        what = ("lexicographic (all)" if lexi else "random")
        workload = (f"{cfg}: {T} taxa x {S} SNPs, {Q} {what} quartets"
                    + (f" in one batch cut over {world} GPUs (strong scaling)" if multi and not args.weak else "")
                    + (f" = {Q // world} per GPU (weak scaling)" if args.weak else "")
                    + f", subsample_snps={sub}, results delivered to host arrays inside the step")
        line = {
            "metric": "resolved quartets/sec (whole node)", "value": value, "unit": "quartets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if (args.weak or not multi) else "strong",
            "vs_baseline": None,
            "dtype": "u8 scan / u32 counts / f64 SVD", "data": "synthetic",
            "config": {"workload": workload, "quartets": Q, "taxa": T, "snps": S, "subsample_snps": sub,
                       "parallelism": (f"quartet-sharded x{world}, one all-gather per result piece, rows to rank 0's host"
                                       if multi else "one GPU, result D2H overlapped inside tq_resolve_to_host")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "note": ("algorithmic bytes (SURVEY 8d: 4*S+48 per quartet) over the dominant kernel's time; the "
                                  "genotype matrix is L2 / Infinity-Cache resident, so this exceeds the HBM peak and is NOT an "
                                  "efficiency figure -- see kernels[].frac for each kernel against the roof that binds it"),
                         "kernel": "tq_scan_wg_kernel", "kernel_ms": dominant_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_copy_measured_GBs": hbm_copy,
                         "scan_stage_ms_per_step": per_pass["order"] + per_pass["scan"],
                         "kernels_note": ("ms of each kernel per step from 3 extra steps of this run with the singular-value "
                                          "chunks serialised on one stream; in the timed region two chunks overlap on two "
                                          "streams (sum of their event spans per step: "
                                          f"{per_pass['bidiag'] + per_pass['bdsqr'] + per_pass['score']:.3f} ms)"),
                         "kernels": kernel_rooflines(kms_serial, launches_serial, q_rank, S, sub, cfg)},
            "flags": {"zero_data": int((flags & 1).sum()), "degenerate": int(((flags & 2) > 0).sum()),
                      "no_convergence": int(((flags & 8) > 0).sum())},
            "commit": git_head(),
        }
        line.update(extra)
        if args.phases in (1, 2):
            line["INVALID_diagnostic_phases"] = args.phases
        if not args.no_cpu and args.phases in (0, 3) and not multi:      # CPU leg: rank 0 at N=1 only
            quartets_np = np.array(quartets_h)
            cb, parity = cpu_baseline(tmparr, tmpmap, quartets_np, sub, rstat, rscor)
            line["cpu_baseline"] = cb
            line["parity_on_cpu_sample"] = parity
            line["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def bench_c5(args, eng, torch, dist, dev, world, rank):
    from tetrad_amd.replicates import ReplicateRunner
    from tetrad_amd import bootstrap, synth
    seqarr, _, spans = synth.make_c5_source()
    T, S0 = seqarr.shape
    Q = args.quartets or 1_000_000
    runner = ReplicateRunner(eng, seqarr, spans, Q, seed=synth.CONFIG_SEEDS["c5"], sampler=args.sampler,
                             pieces=args.pieces or None)
    sub = not args.full
    for _ in range(args.warmup):
        runner.run(1, sub)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.timing_enable(True)
    t0 = time.perf_counter()
    stats = runner.run(args.steps, sub)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms, launches = eng.timing_read_kernels()
    eng.timing_enable(False)
    if world > 1:
        gdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    runner.close()
    # the Generator advanced exactly as the reference's would: the same draws, replayed on a fresh Generator
    rng_check = None
    if rank == 0 and args.sampler == "host" and (args.warmup + args.steps) <= 12:
        chk = np.random.default_rng(synth.CONFIG_SEEDS["c5"])
        for _ in range(args.warmup + args.steps):
            bootstrap.draw_replicate(len(spans), chk)                       # run_inference.py:117-123
            chk.choice(synth.comb(T, 4), size=Q, replace=False)             # combinations.py:113
        rng_check = bool(chk.bit_generator.state == runner.rng_state_after[-1])
    if rank == 0:
        S = int(np.mean(stats["sites"]))
        q_rank = runner.n_local
        A = 4 * S + 48
        per_pass = {k: v / max(1, launches) for k, v in kms.items()}
        bytes_per_launch = q_rank * A + 4 * S
        achieved = bytes_per_launch / (per_pass["scan"] / 1e3) / 1e9
        line = {
            "metric": "resolved quartets/sec (whole node)", "value": Q * args.steps / elapsed, "unit": "quartets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 scan / u32 counts / f64 SVD", "data": "synthetic",
            "config": {"workload": f"c5: {T} taxa x {S0} source SNPs, {args.steps} bootstrap replicates (one per step): device "
                                   f"locus resample + column shuffle + IUPAC resolution, fresh sample of {Q} quartets per "
                                   f"replicate cut over {world} GPU(s), subsample_snps={sub}, rows delivered to rank 0's host",
                       "quartets_per_replicate": Q, "taxa": T, "snps": S0, "mean_replicate_sites": S,
                       "subsample_snps": sub, "sampler": args.sampler,
                       "parallelism": f"replicate-sequential RNG (run_inference.py:378-407), quartet-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "kernel": "tq_scan_wg_kernel", "kernel_ms": per_pass["scan"],
                         "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": None,
                         "traffic_source": "not measured in this run",
                         "all_kernels_ms_per_step": sum(per_pass.values()),
                         "kernels_note": ("event spans of the timed region; the singular-value chunks of a replicate run on two "
                                          "streams, so the bidiag / bdsqr / score spans overlap and over-state each kernel's own "
                                          "duration (the c3 line of the default run has them serialised)"),
                         "kernels": kernel_rooflines(kms, launches, q_rank, S, sub, "c3")},
            "host_draw_ms_per_replicate_mean": float(np.mean(stats["host_ms"])),
            "main_thread_wait_for_draws_ms_mean": float(np.mean(stats["wait_ms"])),
            "flags": stats["flags"], "rng_state_matches_reference_draw_order": rng_check,
            "commit": git_head(),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
