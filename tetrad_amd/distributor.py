"""Quartet dispatch and result gather across the GPUs of one node.

Replaces the reference's ipyparallel load-balanced dispatch
(tetrad/src/run_inference.py:184-251 `distributor`, chunk sizing :73-96) with the
MI355X-native scheme of SURVEY.md section 8e:

  * one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI),
  * the global quartet index space [0,Q) is cut into `world` contiguous ranges, balanced +-1;
    the genotype matrix is replicated (<= 26 MB), so there is no scatter,
  * every rank resolves its range with the HIP engine and ONE all-gather of fixed-size
    32-byte records per quartet (3 x f64 score bit patterns + packed {topology, flags, nsnps}, moved as int64) returns
    the whole batch, in global-index order, to every rank (rank 0 feeds QMC).

The per-rank compute step is injectable so that the sharding / padding / gather logic is
covered by world_size-2 `gloo` tests on CPU; the default is the GPU engine and there is no
CPU fallback in this module.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Iterable, Optional

import numpy as np

RECORD_WORDS = 4        # 3 score bit patterns + 1 packed word (int64 each) -> 32 bytes per quartet


def shard_bounds(Q: int, world: int) -> list[tuple[int, int]]:
    """Contiguous ranges [lo,hi) of the global index space, sizes differ by at most one."""
    base, extra = divmod(int(Q), int(world))
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def get_chunksize(nquartets: int, ncores: int) -> int:
    """Same rule as run_inference.py:73-96 (kept for callers that stream chunks)."""
    breaks = 2
    if nquartets < 5000:
        breaks = 1
    if nquartets > 100000:
        breaks = 8
    if nquartets > 500000:
        breaks = 16
    if nquartets > 5000000:
        breaks = 32
    chunk = nquartets // (breaks * ncores)
    extra = nquartets % (breaks * ncores)
    return max(1, chunk + extra)


def pack_records(rstat: np.ndarray, rscor: np.ndarray, flags: np.ndarray) -> np.ndarray:
    """(rstat u32[n,2], rscor f64[n,3], flags u8[n]) -> int64[n,4] records (scores as raw bits)."""
    n = rstat.shape[0]
    rec = np.empty((n, RECORD_WORDS), dtype=np.int64)
    rec[:, :3] = np.ascontiguousarray(rscor, dtype=np.float64).view(np.int64)
    word = (rstat[:, 0].astype(np.uint64) & np.uint64(0xFF)) | (flags.astype(np.uint64) << np.uint64(8)) \
        | (rstat[:, 1].astype(np.uint64) << np.uint64(32))
    rec[:, 3] = word.view(np.int64)
    return rec


def unpack_records(rec: np.ndarray):
    """Inverse of pack_records."""
    rec = np.ascontiguousarray(rec, dtype=np.int64)
    word = rec[:, 3].copy().view(np.uint64)
    rstat = np.empty((rec.shape[0], 2), dtype=np.uint32)
    rstat[:, 0] = (word & np.uint64(0xFF)).astype(np.uint32)
    rstat[:, 1] = (word >> np.uint64(32)).astype(np.uint32)
    flags = ((word >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
    return rstat, np.ascontiguousarray(rec[:, :3]).view(np.float64), flags


def gpu_compute(device_id: int) -> Callable:
    """Default per-rank compute: the HIP engine on `device_id` (data uploaded once per replicate)."""
    from .resolve_quartets import _ensure_resident, get_engine

    def run(tmparr, tmpmap, quartets, subsample_snps):
        eng = get_engine(device_id)
        _ensure_resident(eng, tmparr, tmpmap)
        return eng.resolve(quartets, subsample_snps)

    return run


def resolve_sharded(tmparr, tmpmap, quartets, subsample_snps: bool = True, *, group=None,
                    compute: Optional[Callable] = None, device=None):
    """Resolve `quartets` (u32[Q,4], identical on every rank) across the ranks of `group`.

    Returns (quartets, rstat u32[Q,2], rscor f64[Q,3], flags u8[Q]) on every rank, rows in the
    order of `quartets`.  With an uninitialised process group this is the 1-GPU path."""
    import torch
    import torch.distributed as dist

    quartets = np.ascontiguousarray(quartets, dtype=np.uint32).reshape(-1, 4)
    Q = quartets.shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        backend = dist.get_backend(group)
    else:
        world, rank, backend = 1, 0, None
    lo, hi = shard_bounds(Q, world)[rank]
    if compute is None and world > 1:
        return _resolve_sharded_device(tmparr, tmpmap, quartets, subsample_snps, group, world, rank, backend,
                                       lo, hi, device)
    if compute is None:
        dev_id = torch.cuda.current_device() if device is None else torch.device(device).index
        compute = gpu_compute(dev_id)
    rstat, rscor, flags = compute(tmparr, tmpmap, quartets[lo:hi], subsample_snps)
    if world == 1:
        return quartets, rstat, rscor, flags
    # fixed-size slabs: pad every rank's records to ceil(Q/world) rows
    slab = -(-Q // world)
    rec = np.zeros((slab, RECORD_WORDS), dtype=np.int64)
    rec[:hi - lo] = pack_records(rstat, rscor, flags)
    tdev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    local = torch.from_numpy(rec).to(tdev)
    gathered = torch.empty((world * slab, RECORD_WORDS), dtype=torch.int64, device=tdev)
    dist.all_gather_into_tensor(gathered, local, group=group)
    g = gathered.cpu().numpy().reshape(world, slab, RECORD_WORDS)
    parts = [g[r, :b - a] for r, (a, b) in enumerate(shard_bounds(Q, world))]
    rstat, rscor, flags = unpack_records(np.concatenate(parts, axis=0))
    return quartets, rstat, rscor, flags


def _resolve_sharded_device(tmparr, tmpmap, quartets, subsample_snps, group, world, rank, backend, lo, hi, device):
    """The N>1 path with the HIP engine: the rank's rows never visit the host before the gather.  The
    engine writes `[rstat u32[slab,2] | rscor f64[slab,3] | flags u8[slab]]` into ONE device slab
    (33 bytes per quartet, padded to ceil(Q/world) rows), one all-gather (RCCL over xGMI with the
    "nccl" backend) collects the slabs of all ranks, one D2H brings them to the host."""
    import torch
    import torch.distributed as dist
    from .resolve_quartets import _ensure_resident, get_engine

    Q = quartets.shape[0]
    dev_id = torch.cuda.current_device() if device is None else torch.device(device).index
    dev = torch.device("cuda", dev_id)
    eng = get_engine(dev_id)
    _ensure_resident(eng, tmparr, tmpmap)
    slab = -(-Q // world)
    n = hi - lo
    blk = (33 * slab + 7) // 8 * 8                   # bytes per rank, 8-byte aligned
    local = torch.zeros(blk, dtype=torch.uint8, device=dev)
    if n:
        d_q = torch.from_numpy(quartets[lo:hi].astype(np.int32)).to(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        eng.resolve_dev(d_q.data_ptr(), n, subsample_snps, local.data_ptr(), local.data_ptr() + 8 * slab,
                        local.data_ptr() + 32 * slab, stream)
    if backend == "nccl":
        gathered = torch.empty(world * blk, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, local, group=group)
        g = gathered.cpu().numpy()
    else:                                           # gloo rehearsal: the collective runs on host tensors
        torch.cuda.synchronize(dev)
        loc = local.cpu()
        gathered = torch.empty(world * blk, dtype=torch.uint8)
        dist.all_gather_into_tensor(gathered, loc, group=group)
        g = gathered.numpy()
    g = g.reshape(world, blk)
    rstat = np.empty((Q, 2), dtype=np.uint32)
    rscor = np.empty((Q, 3), dtype=np.float64)
    flags = np.empty(Q, dtype=np.uint8)
    for r, (a, b) in enumerate(shard_bounds(Q, world)):
        m = b - a
        rstat[a:b] = g[r, :8 * slab].view(np.uint32).reshape(slab, 2)[:m]
        rscor[a:b] = g[r, 8 * slab:32 * slab].view(np.float64).reshape(slab, 3)[:m]
        flags[a:b] = g[r, 32 * slab:33 * slab][:m]
    return quartets, rstat, rscor, flags


# --------------------------------------------------------------------------------------------
# consumer of the triple: the quartets TSV (run_inference.py:232-244)
# --------------------------------------------------------------------------------------------
def format_tsv_bytes(rqrts: np.ndarray, rscor: np.ndarray, rstat: np.ndarray) -> bytes:
    """The rows of the quartets TSV as bytes (native formatter of the C ABI: `tq_format_tsv`)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    q = np.ascontiguousarray(rqrts, dtype=np.uint32).reshape(-1, 4)
    st = np.ascontiguousarray(rstat, dtype=np.uint32).reshape(-1, 2)
    sc = np.ascontiguousarray(rscor, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    if st.shape[0] != n or sc.shape[0] != n:
        raise ValueError("rqrts, rscor and rstat must have the same number of rows")
    cap = 96 * n + 4096
    written = ctypes.c_int64()
    for _ in range(2):
        buf = np.empty(cap, dtype=np.uint8)
        rc = lib.tq_format_tsv(q.ctypes.data, st.ctypes.data, sc.ctypes.data, n, buf.ctypes.data, cap,
                               ctypes.byref(written))
        if rc == 0:
            return buf[:written.value].tobytes()
        if rc != -6:                                               # TQ_ERR_OOM carries the size to retry with
            raise _lib.TetradHipError(rc, "tq_format_tsv")
        cap = written.value
    raise _lib.TetradHipError(rc, "tq_format_tsv: buffer sizing failed")


def format_tsv(rqrts: np.ndarray, rscor: np.ndarray, rstat: np.ndarray) -> str:
    """Same text as the reference's
    ``pd.concat([DataFrame(rqrts), DataFrame(rscor), DataFrame(rstat)], axis=1)
      .to_csv(sep="\\t", float_format='%.6f', index=False, header=False)``
    (run_inference.py:233-234): 9 columns  a b c d score0 score1 score2 topo nsnps."""
    return format_tsv_bytes(rqrts, rscor, rstat).decode("ascii")


def distributor(database_file: Path, qrts_file: Path, nsamples: int, qiter: Iterable, subsample_snps: bool,
                client=None, *, group=None, compute: Optional[Callable] = None) -> Path:
    """Mirror of run_inference.py:184-251: resolve every chunk of `qiter` and append the rows
    to `qrts_file`.  `client` (an ipyparallel Client in the reference) is accepted and ignored:
    the GPUs of the node take its place.  Only rank 0 writes the file."""
    import torch.distributed as dist
    from .resolve_quartets import load_database

    qrts_file = Path(qrts_file)
    rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
    if rank == 0:
        qrts_file.write_text("")                                   # :209
    tmparr, tmpmap = load_database(database_file)
    for chunk in qiter:                                            # :216-223
        qrts = np.array(list(chunk), dtype=np.uint32).reshape(-1, 4)
        rqrts, rstat, rscor, _ = resolve_sharded(tmparr, tmpmap, qrts, subsample_snps, group=group,
                                                 compute=compute)
        if rank == 0:
            with open(qrts_file, "ab") as out:                     # :240-244
                out.write(format_tsv_bytes(rqrts, rscor, rstat))
    return qrts_file
