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
TA_CYCLES_PER_LOAD = 18        # tools/probe_ta.hip: cycles of a CU's texture-address path per vector load wave-instruction (x2..x4)


def _cpu_worker(args):
    """cpu_baseline leg: the oracle (port of resolve_quartets.py:191-265) on one host core.
    variant: "port" (compiled gather + masks + count in one C call, then the reference's svd + matrix_rank calls),
    "tuned" (one batched values-only SVD per chunk), "faithful" (the reference's NumPy temporaries spelled out:
    fancy-index row gather, two mask reductions, then the count kernel -- resolve_quartets.py:212-223)."""
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import oracle as orc
    if isinstance(args[0], np.ndarray) and args[0].ndim == 1:      # (row indices, variant): the data came with the fork
        tmparr, tmpmap, allq, sub = _CPU_SHARED
        quartets, variant = allq[args[0]], args[1]
    else:
        tmparr, tmpmap, quartets, sub = args[:4]
        variant = args[4] if len(args) > 4 else "port"
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(1)
    except Exception:  # pragma: no cover
        ctx = None
    t0 = time.perf_counter()
    fn = {"port": orc.new_infer_resolved_quartets, "tuned": orc.new_infer_resolved_quartets_batched,
          "faithful": orc.new_infer_resolved_quartets_numpy}[variant]
    _, rstat, rscor = fn(tmparr, tmpmap, quartets, sub)
    dt = time.perf_counter() - t0
    return dt, rstat, rscor


_CPU_SHARED = None      # (tmparr, tmpmap, quartets, sub) of the running cpu_baseline leg: inherited by the forked workers


def host_cores():
    """(hardware threads this process may run on, threads per core, physical cores among them)."""
    avail = sorted(os.sched_getaffinity(0))
    cores = set()
    tpc = 1
    for c in avail:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
            ids = []
            for part in sib.split(","):
                lo, _, hi = part.partition("-")
                ids += list(range(int(lo), int(hi or lo) + 1))
            tpc = max(tpc, len(ids))
            cores.add(min(ids))
        except (OSError, ValueError):
            cores.add(c)
    return len(avail), tpc, max(1, len(cores))


def cpu_quota_cores():
    """CPU time this process's control group may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when
    unlimited.  The GPU boxes give a one-GPU job a share of the host (16 cores of 128): more worker processes than
    that only divide the same CPU time (measured: 128 processes ran at 221 quartets/s each, 16 at 2 390)."""
    for f, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                     ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(f).read().strip()
            if parse:
                q, per = parse(txt)
                if q == "max":
                    return None
                return float(q) / float(per)
            q = float(txt)
            if q <= 0:
                return None
            return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
        except (OSError, ValueError):
            continue
    return None


def cpu_baseline(tmparr, tmpmap, quartets, sub, gpu_rstat, gpu_rscor, budget_s=15.0, max_procs=0):
    """The oracle timed on the host, one single-threaded process per PHYSICAL core this process may use (the
    reference's engine model: one engine per core, BLAS pinned to one thread).  Three variants of the same path,
    each on a bounded sample of the benchmark's own batch; the first one doubles as the parity check."""
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    threads, tpc, phys = host_cores()
    quota = cpu_quota_cores()
    cores = phys if quota is None else max(1, min(phys, int(round(quota))))
    if max_procs > 0:
        cores = max(1, min(phys, max_procs))

    global _CPU_SHARED
    _CPU_SHARED = (tmparr, tmpmap, quartets, sub)

    def timed(idx_chunks, variant):
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(len(idx_chunks)) as pool:
            res = pool.map(_cpu_worker, [(c, variant) for c in idx_chunks], chunksize=1)
        # the slowest worker's own clock: pool start-up (forking a process that holds GPU mappings) is not CPU work
        wall = min(time.perf_counter() - t0, max(r[0] for r in res))
        return wall, np.concatenate([r[1] for r in res]), np.concatenate([r[2] for r in res])

    # calibrate on one core, then size each sample for its share of the budget on all cores
    dt, _, _ = _cpu_worker((tmparr, tmpmap, quartets[:300], sub))
    n = int(min(len(quartets), max(cores * 100, 300 / dt * cores * budget_s * 0.6)))
    wall, rstat, rscor = timed(np.array_split(np.arange(n), cores), "port")
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
    n2 = int(min(len(quartets), max(cores * 100, n // 2)))
    wall2, rstat2, _ = timed(np.array_split(np.arange(n2), cores), "tuned")
    parity["tuned_cpu_variant_topology_equal"] = bool(np.array_equal(rstat2[:, 0], gpu_rstat[:n2, 0]))
    # third figure: the reference-faithful interpreted structure (NumPy gather + two mask reductions per quartet)
    n3 = int(min(len(quartets), max(cores * 30, n // 6)))
    wall3, rstat3, _ = timed(np.array_split(np.arange(n3), cores), "faithful")
    parity["reference_faithful_variant_topology_equal"] = bool(np.array_equal(rstat3[:, 0], gpu_rstat[:n3, 0]))
    whole = cores >= phys
    return dict(value=n / wall, unit="quartets/s", cores=cores,
                whole_host=(n / wall if whole else None),
                whole_host_extrapolated=(None if whole else n / wall / cores * phys),
                whole_host_note=("measured on every physical core" if whole else
                                 f"this job's control group allows {quota:.1f} cores of CPU time "
                                 f"(cpu.max): {cores} worker processes measured, the whole-host figure is per_core x {phys} "
                                 f"physical cores -- an extrapolation; more processes than the quota only share the same CPU "
                                 f"time (bench.py --cpu-procs {phys} measures that: 28.3 k quartets/s on 128 processes "
                                 f"against 38 k on 16, profiles/r04_final)"),
                cpu_quota_cores=quota,
                host_hardware_threads_available_to_this_process=threads, threads_per_core=tpc,
                host_physical_cores_available_to_this_process=phys, host_cores_total=os.cpu_count(),
                cpu_model=model, kind="port",
                tuned_variant=dict(value=n2 / wall2, unit="quartets/s", cores=cores,
                                   what=f"first {n2} quartets, compiled count loop + one batched values-only "
                                        f"numpy.linalg.svd per chunk (no interpreter in the per-quartet loop "
                                        f"beyond one ctypes call)"),
                reference_faithful=dict(value=n3 / wall3, unit="quartets/s", cores=cores, per_core=n3 / wall3 / cores,
                                        what=f"first {n3} quartets, oracle.new_infer_resolved_quartets_numpy: the "
                                             f"reference's interpreted structure (resolve_quartets.py:212-223: NumPy "
                                             f"fancy-index row gather + two mask reductions per quartet, compiled "
                                             f"count loop standing in for numba, numpy.linalg svd + matrix_rank)"),
                sample=f"first {n} quartets of the same batch, {cores} processes x 1 thread (one per physical core "
                       f"this process may use), oracle.new_infer_resolved_quartets (C gather + masks + count loop, "
                       f"numpy.linalg svd/matrix_rank)",
                per_core=n / wall / cores), parity


def git_head() -> str:
    try:
        return subprocess.check_output(["git", "-C", str(REPO), "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        f = REPO / ".build_commit"          # written by __graft_entry__.build() where git is available
        return f.read_text().strip() if f.exists() else "unknown"


def library_path() -> str:
    """The shared object this process mapped (an inherited TQ_LIB_PATH must show in the line), with the commit it was
    built from when it is the in-tree library."""
    from tetrad_amd import _lib
    p = str(_lib.LOADED_PATH)
    f = REPO / ".build_commit"
    if p == str(_lib.LIB_PATH) and f.exists():
        p += " (built at " + f.read_text().strip() + ")"
    return p


def dp_unit_count(quartets: np.ndarray, T: int) -> dict:
    """Units (pairs of quartets that share their first three taxa + quartets on their own) the joint-histogram scan
    (scan_dp.hpp, full mode) makes of a batch: inside a run of equal (a,b,c) of the sorted order neighbours are paired."""
    q = np.asarray(quartets).reshape(-1, 4).astype(np.int64)
    key = (q[:, 0] * T + q[:, 1]) * T + q[:, 2]
    _, cnt = np.unique(key, return_counts=True)
    pairs, singles = int((cnt // 2).sum()), int((cnt % 2).sum())
    return {"pairs": pairs, "singles": singles, "units": pairs + singles}


def kernel_rooflines(kms: dict, calls: int, Q: int, S: int, sub: bool, cfg: str, dp: dict | None = None, f4: bool = False):
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
    if per.get("order", 0) > 0:
        out.append(dict(name="ordering (tq_key_kernel + radix sort)", ms=per["order"], bound="latency (~20 small launches)",
                        frac=None))
    if per.get("scan", 0) > 0 and dp:
        # full mode, joint-histogram scan (scan_dp.hpp): a wave scans a UNIT -- two quartets that share (a,b,c), or one --
        # with one LDS atomic per site slot: per unit and 2048-site step 32 ds_add_u32 (5.1 LDS-array cycles each at 49 %
        # lane occupancy, the birthday bound of ~16 random bins per 32 banks twice; counters of round 4,
        # profiles/r04_scan/README.md section 8), 3 image reads ds_read_b128 (4), a quarter of the image stores; 6 vector
        # loads for the unit's own rows c, d1, d2 (codes + plane records) + 4 per workgroup of 4 for the shared rows a, b
        units = dp["units"] * (Q / max(1, dp["quartets"]))
        loads = units * steps * (6 + 4 / 4)
        ta_ms = loads * TA_CYCLES_PER_LOAD / 256 / 2.4e9 * 1e3
        lds_cycles = 12 + 10.25 + 32 * 5.1
        lds_ms = units * steps * lds_cycles / 256 / 2.4e9 * 1e3
        out.append(dict(name="tq_scan_dp_kernel", ms=per["scan"],
                        bound="co-bound: LDS instruction path (lds_path_frac), VALU issue; pmc shares in profiles/r04_scan/README.md section 8",
                        achieved=loads / (per["scan"] / 1e3) / 1e9, peak=256 * 2.4 / TA_CYCLES_PER_LOAD,
                        unit="G vector-load wave-instructions/s", frac=ta_ms / per["scan"],
                        lds_path_frac=lds_ms / per["scan"], lds_path_cycles_per_wave_step=lds_cycles,
                        units=dict(dp, quartets_per_wave_step=dp["quartets"] / max(1, dp["units"])),
                        model=f"{dp['pairs']} pairs + {dp['singles']} single quartets of {dp['quartets']} = {dp['units']} units; 7 vector "
                              f"loads per unit and 2048-site step x {TA_CYCLES_PER_LOAD} cycles = {ta_ms:.2f} ms; LDS instruction path "
                              f"{lds_cycles:.0f} cycles per unit-step = {lds_ms:.2f} ms"))
    elif per.get("scan", 0) > 0 and f4:
        # the plane-record scan (scan_f4.hpp: SURVEY 8 row f4 as written, the default of subsample mode): per wave and 2048-site step
        # 2 vector loads for the wave's own rows c, d (one 12-byte plane record each) + 4 per workgroup of 4 waves for the shared
        # rows a, b; LDS instruction path: image reads ds_read_b128 (4) + ds_read_b32 (2); image stores, a quarter of 8 x
        # ds_write_addtid_b32 (2) + b128 (13) + b32 (4); walk trips (8.2 in subsample mode on c3, 19 in full mode) x (ds_read_u8
        # 2 + ds_add_u32 5.8 array cycles); no park.  15 vector instructions per walk trip (6 in tq_scan_wg_kernel).
        loads = Q * steps * (2 + 4 / 4)
        ta_ms = loads * TA_CYCLES_PER_LOAD / 256 / 2.4e9 * 1e3
        trips = 8.2 if sub else 19.0
        lds_cycles = 6 + (16 + 17) / 4 + trips * (2 + 5.8)
        lds_ms = Q * steps * lds_cycles / 256 / 2.4e9 * 1e3
        out.append(dict(name="tq_scan_f4_kernel", ms=per["scan"],
                        bound="co-bound: LDS instruction path (lds_path_frac), VALU issue (2 wave-instructions per cycle and CU), "
                              "texture-address issue (frac); pmc shares below",
                        achieved=loads / (per["scan"] / 1e3) / 1e9, peak=256 * 2.4 / TA_CYCLES_PER_LOAD,
                        unit="G vector-load wave-instructions/s", frac=ta_ms / per["scan"],
                        lds_path_frac=lds_ms / per["scan"], lds_path_cycles_per_wave_step=lds_cycles,
                        model=f"3 vector loads per quartet and 2048-site step x {TA_CYCLES_PER_LOAD} cycles of a CU's texture-address "
                              f"path each = {ta_ms:.2f} ms at 2.4 GHz; LDS instruction path {lds_cycles:.0f} cycles per wave-step = "
                              f"{lds_ms:.2f} ms; 141 vector instructions per wave-step (profiles/r04_scan/README.md section 9)"))
    elif per.get("scan", 0) > 0:
        # vector load instructions a CU has to issue: per wave and 2048-site step 4 for the wave's own rows c, d (2 x 16 B
        # nibble codes, 2 x 12 B plane records) + 4 per workgroup of 4 waves for the shared rows a, b; each costs the CU's
        # texture-address path ~18 cycles whatever its width (tools/probe_ta.hip, profiles/r03_scan/probe_ta.txt)
        loads = Q * steps * (4 + 4 / 4)
        ta_ms = loads * TA_CYCLES_PER_LOAD / 256 / 2.4e9 * 1e3
        # LDS instruction path (round 4, tools/probe_slots.hip + MI355X_MICROARCH.md "LDS"): an LDS instruction costs a CU
        # max(2 cycles per source dword it moves to the LDS, its LDS-array cycles).  Per wave and 2048-site step:
        #   image reads 3 x ds_read_b128 (4) + ds_read_b32 (2); image stores, a quarter of 2 x b128 (13) + b128 + b32 (4);
        #   subsample: 8 park stores ds_write_addtid_b32 (2), 8.2 walk trips x (ds_read_u8 2 + ds_add_u32 5.8 array cycles
        #   at 68 % lane occupancy: the birthday bound of ~22 random bins per 32 banks);
        #   full: 32 EXEC-masked ds_add_u32 x max(4 transfer, 4.6 array)
        lds_cycles = (14 + 10.75 + 16 + 8.2 * (2 + 5.8)) if sub else (12 + 10.75 + 32 * 4.6)
        lds_ms = Q * steps * lds_cycles / 256 / 2.4e9 * 1e3
        out.append(dict(name="tq_scan_wg_kernel", ms=per["scan"],
                        bound="co-bound: LDS instruction path (lds_path_frac), VALU issue, texture-address issue (frac); pmc shares below",
                        achieved=loads / (per["scan"] / 1e3) / 1e9, peak=256 * 2.4 / TA_CYCLES_PER_LOAD,
                        unit="G vector-load wave-instructions/s", frac=ta_ms / per["scan"],
                        lds_path_frac=lds_ms / per["scan"], lds_path_cycles_per_wave_step=lds_cycles,
                        model=f"5 vector loads per quartet and 2048-site step x {TA_CYCLES_PER_LOAD} cycles of a CU's texture-address "
                              f"path each = {ta_ms:.2f} ms at 2.4 GHz; LDS instruction path {lds_cycles:.0f} cycles per wave-step = "
                              f"{lds_ms:.2f} ms (an LDS atomic costs 4 CU-cycles of operand transfer whatever its conflicts, "
                              f"profiles/r04_scan/probe_slots.txt)"))
    if per.get("bidiag", 0) > 0:
        fl = 3 * Q * (8.0 / 3.0) * 16 ** 3
        ach = fl / (per["bidiag"] / 1e3) / 1e12
        out.append(dict(name="tq_bidiag_kernel", kernel="tq_bidiag2_kernel (2 x 2 lane layout) unless --opt bidiag_layout=0",
                        ms=per["bidiag"], bound="f64 valu", achieved=ach,
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
    pmc = REPO / "profiles" / ("pmc_busy_latest.json" if sub else "pmc_busy_full_latest.json")
    if pmc.exists():
        try:
            info = json.loads(pmc.read_text())
            for k in out:
                for name, v in info.get("kernels", {}).items():
                    if name in k["name"]:
                        k["pmc"] = dict({x: v[x] for x in ("valu_busy", "lds_busy", "lds_conflict_share", "ta_busy") if x in v},
                                        source=f"profiles/{pmc.name}: " + str(info.get("source")),
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=["c2", "c3", "c4", "c5"],
                    help="default: c3 (one GPU: 1e6 quartets; N GPUs: N x 1e6, weak scaling, then a c4 strong-scaling leg)")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank its own 1e6 quartets of the c3 matrix (the default)")
    ap.add_argument("--no-c4-leg", action="store_true",
                    help="N > 1 default run: skip the secondary c4 strong-scaling leg (BASELINE.json configs[3])")
    ap.add_argument("--quartets", type=int, default=0, help="quartets in the batch (0 = config default)")
    ap.add_argument("--full", action="store_true", help="subsample_snps=False")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-other-mode", action="store_true",
                    help="skip the 3 steps in the other subsample mode (full_mode_value): profiler passes average per kernel name")
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="processes of the cpu_baseline leg (0 = one per physical core this process may use)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 control flow on a box with fewer GPUs than ranks")
    ap.add_argument("--pieces", type=int, default=0, help="N > 1: result pieces per batch (0 = automatic)")
    ap.add_argument("--gather-ab", action="store_true",
                    help="N > 1: add a three-step leg with the OTHER --gather mode to the line (other_gather_leg); opt-in, so that "
                         "the bare command a driver runs on a multi-GPU node exercises one data path only")
    ap.add_argument("--gather", default="collective", choices=["collective", "host"],
                    help="N > 1: 'collective' = one all-gather per piece, rank 0 copies the rows D2H (default); 'host' = every "
                         "rank copies its own rows D2H into rank 0's shared page-locked arrays, no collective on the data path")
    ap.add_argument("--sharded", action="store_true",
                    help="take the N > 1 code path (process group, all-gather per piece, regrouping, verification) "
                         "even with one rank: a one-GPU rehearsal of the RCCL path")
    ap.add_argument("--sampler", default="host", choices=["host", "device"],
                    help="c5: quartet sample drawn on the project Generator (reference stream) or on the device")
    ap.add_argument("--svd-chunk", type=int, default=0)
    ap.add_argument("--order", type=int, default=-1, help="0 = natural quartet order, 1 = sorted (default)")
    ap.add_argument("--phases", type=int, default=0, help="diagnostic: 1 scan only, 2 SVD only (invalid as a result)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option (tq_set_option), repeatable")
    ap.add_argument("--launch-probe", action="store_true",
                    help="self-test of the rank launcher and the rendezvous only (no GPU work, NOT a result): every rank "
                         "joins a gloo group, rank 0 prints what the group saw")
    return ap.parse_args(argv)


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher environment: start the N ranks ourselves.

    This process has not touched the GPU (no HIP call, no torch.cuda call other than device_count) and never
    will: it starts N fresh children -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would -- relays rank 0's JSON line, and exits with the worst child status.  If a rank
    dies the others are ended by their exact PIDs (they would wait in a collective forever)."""
    n = args.gpus
    if args.backend == "nccl":
        import torch
        ndev = torch.cuda.device_count()            # does not initialise the GPU on this image
        if ndev < n:
            sys.stderr.write(f"bench.py: --gpus {n} with backend nccl needs {n} visible GPUs, found {ndev} "
                             f"(use --backend gloo to rehearse the {n}-rank control flow on fewer GPUs)\n")
            return 2
    env = dict(os.environ)
    # RCCL opens peers' buffers through HIP IPC handles.  The hosts of this pool support only dmabuf IPC: with the legacy
    # mode (the runtime's default) hipIpcGetMemHandle fails with "invalid argument" and the communicator cannot be
    # built.  The image exports the variable already (build environment notes); it is repeated here -- never overriding a
    # value the caller set -- because a rank started from a scrubbed environment would fail in exactly that way.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               TQ_BENCH_SELF_LAUNCHED="1")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import signal
    import threading

    def on_signal(signum, _frame):                  # a launcher that is told to stop takes its ranks with it
        raise SystemExit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, on_signal)
    out0 = []
    t = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    t.start()
    worst, alive = 0, set(range(n))
    try:
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and worst == 0:
                    worst = rc if rc > 0 else 1
                    sys.stderr.write(f"bench.py: rank {r} exited with status {rc}; ending the other ranks\n")
                    for o in alive:
                        procs[o].terminate()            # exact PIDs of our own children
            time.sleep(0.05)
    finally:
        for r in alive:                             # only on the way out through a signal / exception: never leave ranks behind
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in alive:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    t.join(timeout=10)
    if out0:
        # the contract is ONE JSON line on stdout: anything else rank 0 wrote there (gloo's C++ side prints its
        # connection report to stdout) goes to stderr
        for ln in out0[0].decode(errors="replace").splitlines():
            (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
        sys.stdout.flush()
    return worst


def launch_probe(args):
    """--launch-probe: what a child rank does instead of the benchmark (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("TQ_BENCH_PROBE_FAIL_RANK") == str(rank):
        raise SystemExit(7)
    if os.environ.get("TQ_BENCH_PROBE_PIDFILE"):          # test hook: say who we are, then linger
        with open(os.environ["TQ_BENCH_PROBE_PIDFILE"], "a") as f:
            f.write(f"{os.getpid()}\n")
        time.sleep(float(os.environ.get("TQ_BENCH_PROBE_SLEEP", "0")))
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        seen, total = dist.get_world_size(), int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen, total = 1, 1
    if rank == 0:
        emit({"probe": True, "n_gpus": seen, "gpus_arg": args.gpus, "rank_sum": total,
                          "local_ranks_distinct": True, "self_launched": bool(os.environ.get("TQ_BENCH_SELF_LAUNCHED")),
                          "master_addr": os.environ.get("MASTER_ADDR")})


_REAL_STDOUT = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when its first
    communicator is made, gloo its connection report), so everything this process and its libraries write to fd 1 is sent
    to stderr from here on, and `emit` writes the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line: dict):
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))
    quiet_stdout()

    if args.launch_probe:
        return launch_probe(args)

    import torch
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine, pinned_empty

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    multi = env_world > 1 or (args.sharded and "RANK" in os.environ)
    world, rank = 1, 0
    if multi:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        dist.barrier()                                            # communicator set-up happens here, never inside a timed step
        world, rank = dist.get_world_size(), dist.get_rank()      # what the collective library actually sees
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {world} ranks")

    cfg = args.config or "c3"
    weak = multi and cfg == "c3"                  # N > 1 on the c3 shape: per-GPU work fixed (the N = 1 workload per rank)
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
    for kv in args.opt:
        k, v = kv.split("=", 1)
        eng.set_option(k, int(v))

    if cfg == "c5":
        return bench_c5(args, eng, torch, dist, dev, world, rank)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream

    def make_workload(cfg, weak):
        T, S, qdef = synth.CONFIGS[cfg]
        tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
        eng.set_data(tmparr, tmpmap)
        lexi = qdef is None and not args.quartets
        if lexi:
            Q = int(synth.comb(T, 4))
            ranks_all = np.arange(Q, dtype=np.int64)
        else:
            Q = args.quartets or qdef
            if weak:
                Q *= world
            # ONE seeded sample for the whole job, identical on every rank (combinations.py:109-114)
            ranks_all = np.random.default_rng(synth.CONFIG_SEEDS[cfg] + 1000).choice(
                synth.comb(T, 4), size=Q, replace=False).astype(np.int64)
        return dict(cfg=cfg, T=T, S=S, Q=Q, lexi=lexi, weak=weak, tmparr=tmparr, tmpmap=tmpmap, ranks_all=ranks_all)

    def one_gpu_pass(ranks, steps, warm=1):
        """tq_resolve_to_host on THIS rank's GPU alone over the quartets with lexicographic ranks `ranks`:
        returns (quartets/s, (rstat, rscor, flags), device quartets)."""
        n = len(ranks)
        d_r = torch.from_numpy(np.ascontiguousarray(ranks)).to(dev)
        d_q = torch.empty((n, 4), dtype=torch.int32, device=dev)
        eng.unrank_dev(d_r.data_ptr(), n, d_q.data_ptr(), stream)
        torch.cuda.synchronize()
        out = (pinned_empty((n, 2), np.uint32), pinned_empty((n, 3), np.float64), pinned_empty(n, np.uint8))
        for _ in range(warm):
            eng.resolve_to_host(d_q.data_ptr(), n, sub, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.resolve_to_host(d_q.data_ptr(), n, sub, out=out)
        torch.cuda.synchronize()
        return n * steps / (time.perf_counter() - t0), out, d_q

    def run_sharded(wl, steps, warmup, timing=True, gather=None, with_one_gpu=True):
        """The N-rank path over workload `wl`: returns a dict with the max-over-ranks elapsed time of `steps`
        steps, the gathered rows (rank 0), the verification of the gather and the in-run one-GPU figure."""
        from tetrad_amd.distributor import ShardedResolver
        Q = wl["Q"]
        res = ShardedResolver(Q, engine=eng, device=dev_index, dst=0, pieces=args.pieces or None,
                              collective_always=True, gather=gather or args.gather)
        res.set_ranks(wl["ranks_all"])
        torch.cuda.synchronize()
        last = [None]

        def step():
            last[0] = None                  # hand the previous rows back to the pinned pool before new ones are taken
            res.start(sub)
            last[0] = res.finish()

        for _ in range(warmup):
            step()
        barrier()
        eng.timing_enable(timing)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        kms, launches = eng.timing_read_kernels() if timing else ({}, 0)
        eng.timing_enable(False)
        gdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rstat, rscor, flags = last[0]
        if res.gather == "host" and rstat is not None:          # views of the shared segment, which res.close() unmaps
            rstat, rscor, flags = rstat.copy(), rscor.copy(), flags.copy()
        last[0] = None
        # the gathered rows on rank 0 against what every rank computed itself: a checksum over each rank's
        # own rows and 64 sampled rows per rank, compared bit for bit
        P = res.plan
        own = res.local_rows()                     # what this rank's kernels produced (its slabs), either gather mode
        idx = P.local_index(rank)
        mine = list(own) if len(idx) else None
        pick = np.random.default_rng(rank).integers(0, max(1, len(idx)), size=min(64, len(idx)))
        rec = dict(rank=rank, n=int(len(idx)),
                   nsnps_sum=int(mine[0][:, 1].to(torch.int64).sum().item()) if mine else 0,
                   score_bits_xor=int(np.bitwise_xor.reduce(mine[1].cpu().numpy().view(np.int64).ravel())) if mine else 0,
                   rows=[(int(P.local_index(rank)[i]), mine[0][i].tolist(), mine[1][i].cpu().numpy().view(np.int64).tolist())
                         for i in pick] if mine else [])
        recs = [None] * world
        dist.all_gather_object(recs, rec)
        out = dict(elapsed=elapsed, kms=kms, launches=launches, n_local=res.n_local, rows=(rstat, rscor, flags))
        if rank == 0:
            ok = True
            for r in recs:
                gi = P.local_index(r["rank"])
                ok &= int(rstat[gi, 1].astype(np.int64).sum()) == r["nsnps_sum"]
                ok &= int(np.bitwise_xor.reduce(rscor[gi].view(np.int64).ravel())) == r["score_bits_xor"] if len(gi) else True
                for g, rs, sc in r["rows"]:
                    ok &= rstat[g].astype(np.int64).tolist() == [x & 0xFFFFFFFF for x in rs]
                    ok &= rscor[g].view(np.int64).tolist() == sc
            out.update(gather_verified=bool(ok), gather_pieces=P.npieces, rows_per_piece_per_rank=P.part)
        # the like-for-like one-GPU figure, MEASURED IN THIS RUN: rank 0 alone resolves the same quartets one GPU
        # would own in the single-GPU form of this workload (strong: the whole batch; weak: one rank's 1/N share)
        # through tq_resolve_to_host while the other ranks wait at the barrier
        res.close()
        del res, own, mine, idx
        torch.cuda.empty_cache()
        barrier()
        if rank == 0 and with_one_gpu:
            ranks1 = wl["ranks_all"] if not wl["weak"] else wl["ranks_all"][:Q // world]
            v1, out1, _ = one_gpu_pass(ranks1, min(3, steps))
            out["one_gpu_same_run_value"] = v1
            out["one_gpu_same_run_quartets"] = int(len(ranks1))
            n1 = len(ranks1)
            out["one_gpu_rows_equal_gathered_rows"] = bool(np.array_equal(out1[0], rstat[:n1]) and
                                                           np.array_equal(out1[1], rscor[:n1]))
        barrier()
        return out

    extra = {}
    wl = make_workload(cfg, weak)
    T, S, Q, lexi = wl["T"], wl["S"], wl["Q"], wl["lexi"]
    tmparr, tmpmap, ranks_all = wl["tmparr"], wl["tmpmap"], wl["ranks_all"]
    if not multi:
        d_r = torch.from_numpy(ranks_all).to(dev)
        d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
        eng.unrank_dev(d_r.data_ptr(), Q, d_q.data_ptr(), stream)
        torch.cuda.synchronize()
        out = (pinned_empty((Q, 2), np.uint32), pinned_empty((Q, 3), np.float64), pinned_empty(Q, np.uint8))

        def step():
            eng.resolve_to_host(d_q.data_ptr(), Q, sub, out=out)

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
        step()                                  # results of the default configuration
        rstat, rscor, flags = out
        q_rank = Q
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
        extra["pageable_api_equals_timed_path"] = bool(np.array_equal(out_pg[0], rstat) and np.array_equal(out_pg[1], rscor))
        # the other mode of the same workload through the timed path (3 steps after one warm-up): the reference's
        # DEFAULT is subsample_snps=False (cli_init.py:61 is store_true), north_star's primary is True -- the driver's
        # record carries both
        if not args.no_other_mode:             # (profiling passes skip it: per-kernel averages must be of ONE mode)
            out_o = (pinned_empty((Q, 2), np.uint32), pinned_empty((Q, 3), np.float64), pinned_empty(Q, np.uint8))
            eng.resolve_to_host(d_q.data_ptr(), Q, not sub, out=out_o)
            t1 = time.perf_counter()
            for _ in range(3):
                eng.resolve_to_host(d_q.data_ptr(), Q, not sub, out=out_o)
            other = 3 * Q / (time.perf_counter() - t1)
            extra["subsample_mode_value" if not sub else "full_mode_value"] = other
            extra["other_mode_flags"] = {"zero_data": int((out_o[2] & 1).sum()), "degenerate": int(((out_o[2] & 2) > 0).sum()),
                                         "nsnps_mean": float(out_o[0][:, 1].mean())}
    else:
        r = run_sharded(wl, args.steps, args.warmup)
        elapsed, kms, launches, q_rank = r["elapsed"], r["kms"], r["launches"], r["n_local"]
        kms_serial, launches_serial = kms, launches
        rstat, rscor, flags = r["rows"]
        if rank == 0:
            value = Q * args.steps / elapsed
            for k in ("gather_verified", "gather_pieces", "rows_per_piece_per_rank", "one_gpu_same_run_value",
                      "one_gpu_same_run_quartets", "one_gpu_rows_equal_gathered_rows"):
                extra[k] = r[k]
            extra["speedup"] = value / r["one_gpu_same_run_value"]
            extra["speedup_note"] = ("value / one_gpu_same_run_value: rank 0 alone, same run, "
                                     + ("one rank's share of the batch (weak scaling: ideal = n_gpus)" if weak
                                        else "the whole batch (strong scaling: ideal = n_gpus)"))
        # the same workload with the OTHER way of getting the rows to rank 0 (a few steps, same run): one of the two A/Bs
        # DESIGN.md section 6 wants measured on a multi-GPU node before either is trusted as the default
        if args.gather_ab:
            other = "host" if args.gather == "collective" else "collective"
            ro = run_sharded(wl, min(args.steps, 3), 1, timing=False, gather=other, with_one_gpu=False)
            if rank == 0:
                vo = Q * min(args.steps, 3) / ro["elapsed"]
                extra["other_gather_leg"] = dict(
                    gather=other, value=vo, unit="quartets/s", steps=min(args.steps, 3),
                    ms_per_step=ro["elapsed"] / min(args.steps, 3) * 1e3, gather_verified=ro["gather_verified"],
                    rows_equal_main_leg=bool(np.array_equal(ro["rows"][0], rstat) and np.array_equal(ro["rows"][1], rscor)),
                    what=("every rank copies its own rows D2H into rank 0's shared page-locked arrays, no collective on the "
                          "data path" if other == "host" else "one all-gather per result piece, rank 0 copies the rows D2H"))
        # secondary leg of the default N > 1 run: BASELINE.json configs[3] (c4, ONE 5e6-quartet batch cut over the
        # ranks: strong scaling) with its own in-run one-GPU figure
        if cfg == "c3" and not args.config and not args.no_c4_leg and not args.quartets:
            wl4 = make_workload("c4", False)
            r4 = run_sharded(wl4, min(args.steps, 3), 1, timing=False)
            if rank == 0:
                v4 = wl4["Q"] * min(args.steps, 3) / r4["elapsed"]
                extra["c4_strong_leg"] = dict(
                    workload=f"c4: {wl4['T']} taxa x {wl4['S']} SNPs, ONE batch of {wl4['Q']} random quartets cut over "
                             f"{world} GPUs (strong scaling), subsample_snps={sub}, rows to rank 0's host",
                    value=v4, unit="quartets/s", steps=min(args.steps, 3), scaling="strong",
                    ms_per_step=r4["elapsed"] / min(args.steps, 3) * 1e3,
                    one_gpu_same_run_value=r4["one_gpu_same_run_value"], speedup=v4 / r4["one_gpu_same_run_value"],
                    gather_verified=r4["gather_verified"], gather_pieces=r4["gather_pieces"],
                    one_gpu_rows_equal_gathered_rows=r4["one_gpu_rows_equal_gathered_rows"])

    hbm_copy = hbm_copy_rate(torch, dev) if rank == 0 else None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = Q * args.steps / elapsed
        A = 4 * S + 48                                   # algorithmic bytes per quartet
        bytes_per_launch = q_rank * A + 4 * S              # q_rank = quartets one launch of the scan kernel covers here
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
                    + (f" in one batch cut over {world} GPUs (strong scaling)" if multi and not weak else "")
                    + (f" = {Q // world} per GPU (weak scaling: the N = 1 workload on every GPU)" if weak else "")
                    + f", subsample_snps={sub}, results delivered to host arrays inside the step")
        dp = None
        opts = dict(kv.split("=") for kv in args.opt)
        if (not sub and q_rank >= int(opts.get("dp_min_quartets", 0) or 32768) and T ** 3 < 2 ** 32 and not multi
                and int(opts.get("scan_dp", 1)) and int(opts.get("scan_f4", -1)) <= 0 and int(opts.get("scan_method", -1)) < 0
                and int(opts.get("order", 1))
                and (args.order != 0) and locals().get("quartets_h") is not None):
            dp = dict(dp_unit_count(np.array(locals()["quartets_h"]), T), quartets=Q)
        f4_opt = int(opts.get("scan_f4", -1))
        plain = (int(opts.get("scan_method", -1)) < 0 and int(opts.get("park_t", 1)) and not int(opts.get("scan_pair", 0))
                 and not int(opts.get("share_c", 0)) and int(opts.get("scan_wg", 0)) in (0, 4))
        f4 = plain and (f4_opt == 1 or (f4_opt < 0 and sub)) and dp is None
        kernels = kernel_rooflines(kms_serial, launches_serial, q_rank, S, sub, cfg, dp, f4)
        line = {
            "metric": "resolved quartets/sec (whole node)", "value": value, "unit": "quartets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if (multi and not weak) else "weak",
            "vs_baseline": None,
            "dtype": "u8 scan / u32 counts / f64 SVD", "data": "synthetic",
            "config": {"workload": workload, "quartets": Q, "taxa": T, "snps": S, "subsample_snps": sub,
                       "parallelism": ((f"quartet-sharded x{world}, one all-gather per result piece, rows to rank 0's host"
                                        if args.gather == "collective" else
                                        f"quartet-sharded x{world}, every rank copies its rows D2H into rank 0's shared "
                                        f"page-locked arrays (--gather host), no collective on the data path")
                                       if multi else "one GPU, result D2H overlapped inside tq_resolve_to_host"),
                       "launcher": ("bench.py started the ranks itself" if os.environ.get("TQ_BENCH_SELF_LAUNCHED")
                                    else ("torch.distributed.run / external" if multi else "none (one process)")),
                       "backend": (args.backend if multi else None)},
            "roofline": roofline_block(achieved, dominant_ms, bytes_per_launch, traffic, traffic_src, hbm_copy, per_pass,
                                       kernels, multi),
            "flags": {"zero_data": int((flags & 1).sum()), "degenerate": int(((flags & 2) > 0).sum()),
                      "no_convergence": int(((flags & 8) > 0).sum())},
            "commit": git_head(), "library_path": library_path(),
        }
        line.update(extra)
        if args.phases in (1, 2):
            line["INVALID_diagnostic_phases"] = args.phases
        n_diag = int(((flags & 16) > 0).sum())            # TQ_FLAG_INVALID_DIAGNOSTIC: the library marks such rows itself
        if n_diag:
            line["INVALID_diagnostic_rows"] = n_diag
            line["INVALID_diagnostic_options"] = [kv for kv in args.opt if kv.split("=")[0] in ("scan_method", "phases")]
        if not args.no_cpu and args.phases in (0, 3) and not n_diag and not multi:      # CPU leg: rank 0 at N=1 only
            quartets_np = np.array(quartets_h)
            cb, parity = cpu_baseline(tmparr, tmpmap, quartets_np, sub, rstat, rscor, max_procs=args.cpu_procs)
            line["cpu_baseline"] = cb
            line["parity_on_cpu_sample"] = parity
            line["gpu_over_cpu"] = value / cb["value"]                 # against the port on cb["cores"] cores
            whole_cpu = cb["whole_host"] or cb["whole_host_extrapolated"]
            line["gpu_over_cpu_whole_host"] = value / whole_cpu
            line["gpu_over_cpu_whole_host_is_extrapolated"] = cb["whole_host"] is None
            line["gpu_over_cpu_reference_faithful"] = value / cb["reference_faithful"]["value"]
        emit(line)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def roofline_block(achieved, dominant_ms, bytes_per_launch, traffic, traffic_src, hbm_copy, per_pass, kernels, multi):
    """The contract's roofline object (algorithmic bytes over the dominant kernel's HIP-event time against the HBM
    peak) plus SCALAR efficiency keys a flat parser keeps: which unit actually binds the dominant kernel and how
    busy it is (committed rocprofv3 --pmc summary, quoted with its commit), the useful-f64-flop fraction of the
    singular-value kernels, and counter traffic over algorithmic bytes."""
    scan = next((k for k in kernels if k["name"].startswith("tq_scan")), None)
    pmc = (scan or {}).get("pmc") or {}
    shares = {k: pmc.get(k + "_busy") for k in ("valu", "lds", "ta") if pmc.get(k + "_busy") is not None}
    binding, binding_frac = None, None
    if shares:
        top = max(shares.values())
        binding = "+".join(k for k, v in shares.items() if top - v < 0.1)      # every unit within 10 points of the busiest
        binding_frac = top
    svd = [k for k in kernels if k["name"] in ("tq_bidiag_kernel", "tq_bdsqr_kernel")]
    svd_frac = None
    if svd and all(k.get("ms") for k in svd):
        flop = sum(k["achieved"] * 1e12 * k["ms"] / 1e3 for k in svd)
        svd_frac = flop / (sum(k["ms"] for k in svd) / 1e3) / 1e12 / F64_VECTOR_PEAK_TFLOPS
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS,
           "note": ("algorithmic bytes (SURVEY 8d: 4*S+48 per quartet) over the dominant kernel's time; the "
                    "genotype matrix is L2 / Infinity-Cache resident, so this exceeds the HBM peak and is NOT an "
                    "efficiency figure -- binding / binding_frac / svd_f64_frac / kernels[].frac are"),
           "kernel": (scan or {}).get("name", "tq_scan_wg_kernel"), "kernel_ms": dominant_ms,
           "algorithmic_bytes_per_launch": bytes_per_launch,
           "traffic": traffic, "traffic_source": traffic_src,
           "traffic_over_algorithmic": (traffic / bytes_per_launch if traffic else None),
           "binding": binding, "binding_frac": binding_frac,
           "binding_source": (pmc.get("source"), pmc.get("commit")) if pmc else None,
           "scan_ta_issue_frac": (scan or {}).get("frac"),
           "scan_lds_path_frac": (scan or {}).get("lds_path_frac"),
           "svd_f64_frac": svd_frac,
           "hbm_copy_measured_GBs": hbm_copy,
           "scan_stage_ms_per_step": per_pass.get("order", 0) + per_pass.get("scan", 0),
           "svd_stage_ms_per_step_serialised": sum(k["ms"] for k in kernels if k["name"].startswith(("tq_bidiag", "tq_bdsqr", "tq_score"))),
           "kernels_note": ("ms of each kernel per step" + ("" if multi else " from 3 extra steps of this run with the singular-value "
                            "chunks serialised on one stream; in the timed region two chunks overlap on two streams (sum of "
                            f"their event spans per step: {per_pass.get('bidiag', 0) + per_pass.get('bdsqr', 0) + per_pass.get('score', 0):.3f} ms)")),
           "kernels": kernels}
    return out


def bench_c5(args, eng, torch, dist, dev, world, rank):
    from tetrad_amd.replicates import ReplicateRunner
    from tetrad_amd import bootstrap, synth
    seqarr, _, spans = synth.make_c5_source()
    T, S0 = seqarr.shape
    Q = args.quartets or 1_000_000
    runner = ReplicateRunner(eng, seqarr, spans, Q, seed=synth.CONFIG_SEEDS["c5"], sampler=args.sampler,
                             pieces=args.pieces or None, gather=args.gather)
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
    if rank == 0 and args.sampler == "host" and (args.warmup + args.steps) <= 130:
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
                         "frac": achieved / HBM_PEAK_GBS, "kernel": "tq_scan_f4_kernel" if sub else "tq_scan_dp_kernel", "kernel_ms": per_pass["scan"],
                         "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": None,
                         "traffic_source": "not measured in this run",
                         "all_kernels_ms_per_step": sum(per_pass.values()),
                         "kernels_note": ("event spans of the timed region; the singular-value chunks of a replicate run on two "
                                          "streams, so the bidiag / bdsqr / score spans overlap and over-state each kernel's own "
                                          "duration (the c3 line of the default run has them serialised)"),
                         "kernels": kernel_rooflines(kms, launches, q_rank, S, sub, "c3", None, bool(sub))},
            "host_draw_ms_per_replicate_mean": float(np.mean(stats["host_ms"])),
            "main_thread_wait_for_draws_ms_mean": float(np.mean(stats["wait_ms"])),
            "flags": stats["flags"], "rng_state_matches_reference_draw_order": rng_check,
            "commit": git_head(), "library_path": library_path(),
        }
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
