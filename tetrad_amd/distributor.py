"""Quartet dispatch and result gather across the GPUs of one node.

Replaces the reference's ipyparallel load-balanced dispatch
(tetrad/src/run_inference.py:184-251 `distributor`, chunk sizing :73-96) with the
MI355X-native scheme of SURVEY.md section 8e:

  * one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI),
  * the genotype matrix is replicated (<= 26 MB), so there is no scatter; the quartets of one
    replicate are cut into a few large PIECES (contiguous ranges of the global index space), each
    piece into `world` contiguous per-rank parts, balanced +-1,
  * every rank scans ALL its parts in one pass (one ordering + one site-scan launch), then runs the
    singular-value stage piece by piece; as soon as a piece is done its fixed-size slab
    `[rstat u32[p,2] | rscor f64[p,3] | flags u8[p]]` (33 bytes per quartet) goes into ONE all-gather
    per piece, the gathered parts are regrouped into global order on the device and copied to the
    destination rank's host arrays -- all of that under the kernels of the next piece.  A replicate
    of up to ~3e5 quartets per rank is a single piece = one collective per replicate; large batches
    use up to 8 pieces of >= 300k quartets per rank so that only the last piece's gather + D2H is
    exposed.

The per-rank compute step is injectable so that the partition / gather / regroup logic is
covered by world_size-2 `gloo` tests on CPU; the default is the GPU engine and there is no
CPU fallback in this module.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Iterable, Optional

import os

import numpy as np

MIN_PART_ROWS = 300_000        # a rank's part of a piece is at least this many quartets: stage_svd cuts a part into two half-chunks
                               # on two streams, and chunks under ~150k quartets pay the tails of their kernels (a wave per 64
                               # matrices, data-dependent sweep counts).  Measured on one rank through the sharded path, 1e6 quartets
                               # of c3 (tools/pieces_sweep.sh): 1 / 2 / 3 / 4 / 7 pieces = 92.5 / 93.9 / 94.3 / 93.3 / 84.2 M quartets/s;
                               # more pieces hide more of the gather + D2H (which one rank cannot show), so: as many as stay >= 300k on average; with the
                               # decreasing piece sizes of piece_bounds the same sweep reads 93.5 / 96.2 / 95.8 / 93.0 / 89.9
MAX_PIECES = 8


def shard_bounds(Q: int, world: int) -> list[tuple[int, int]]:
    """Contiguous ranges [lo,hi) of the global index space, sizes differ by at most one."""
    base, extra = divmod(int(Q), int(world))
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


PIECE_RATIO = 0.75             # piece i+1 holds this fraction of piece i


def piece_bounds(Q: int, npieces: int) -> list[tuple[int, int]]:
    """Contiguous pieces of [0,Q) of DECREASING size (ratio PIECE_RATIO).  The gather + regroup + D2H of a piece runs
    under the kernels of the next one, so what a step cannot hide is the last piece's: with equal pieces that is 1/n of
    all result traffic (on the destination rank of an 8-GPU job 264 MB per 8e6 quartets over one PCIe link), with
    pieces 1 : 0.75 : 0.56 ... the tail shrinks (3 pieces: 24 % instead of 33 %) while the large early pieces keep the
    singular-value kernels at their efficient chunk sizes."""
    Q, npieces = int(Q), int(npieces)
    if npieces <= 1 or Q <= npieces:
        return shard_bounds(Q, max(1, npieces))
    w = [PIECE_RATIO ** i for i in range(npieces)]
    tot = sum(w)
    sizes = [max(1, int(Q * x / tot)) for x in w]
    sizes[0] += Q - sum(sizes)                        # rounding remainder (and the max(1, .) corrections) to the first piece
    if sizes[0] < 1:                                  # only for tiny Q: fall back to equal pieces
        return shard_bounds(Q, npieces)
    out, lo = [], 0
    for n in sizes:
        out.append((lo, lo + n))
        lo += n
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


class ShardPlan:
    """Partition of the global index space [0,Q): `npieces` contiguous pieces, piece i = [start[i], end[i]),
    cut into `world` contiguous parts of `part[i]` = ceil(len/world) rows (the last parts may be short or
    empty).  The gathered slab of piece i holds world*part[i] >= len rows in global order; the surplus
    rows land beyond end[i] and are overwritten by piece i+1 (or fall into the `world` slack rows at the
    end of the destination arrays)."""

    def __init__(self, Q: int, world: int, pieces: Optional[int] = None):
        self.Q, self.world = int(Q), int(world)
        if pieces is None:
            pieces = min(MAX_PIECES, self.Q // (self.world * MIN_PART_ROWS))
            # the pieces shrink geometrically (piece_bounds): keep the LAST one's per-rank part above ~100k quartets
            while pieces > 1 and (self.Q / self.world) * PIECE_RATIO ** (pieces - 1) / sum(
                    PIECE_RATIO ** i for i in range(pieces)) < 100_000:
                pieces -= 1
        self.npieces = int(max(1, min(MAX_PIECES, pieces, max(1, self.Q))))
        b = piece_bounds(self.Q, self.npieces)
        self.start = [lo for lo, _ in b]
        self.end = [hi for _, hi in b]
        self.part = [max(1, -(-(hi - lo) // self.world)) for lo, hi in b]
        self.rows_padded = self.Q + self.world            # destination arrays: Q rows + slack

    def part_range(self, i: int, rank: int) -> tuple[int, int]:
        lo = min(self.start[i] + rank * self.part[i], self.end[i])
        hi = min(lo + self.part[i], self.end[i])
        return lo, hi

    def local_ranges(self, rank: int) -> list[tuple[int, int]]:
        return [self.part_range(i, rank) for i in range(self.npieces)]

    def local_index(self, rank: int) -> np.ndarray:
        """Global row numbers of the rank's quartets, in the order the rank processes them."""
        r = self.local_ranges(rank)
        return np.concatenate([np.arange(lo, hi, dtype=np.int64) for lo, hi in r]) if r else np.zeros(0, np.int64)

    def slab_bytes(self, i: int) -> int:
        return (33 * self.part[i] + 15) // 16 * 16


def gpu_compute(device_id: int) -> Callable:
    """Single-rank compute through the host-buffer API of the HIP engine on `device_id`."""
    from .resolve_quartets import _ensure_resident, get_engine

    def run(tmparr, tmpmap, quartets, subsample_snps):
        eng = get_engine(device_id)
        _ensure_resident(eng, tmparr, tmpmap)
        return eng.resolve(quartets, subsample_snps)

    return run


class ShardedResolver:
    """Buffers and control flow of the multi-GPU path for batches of Q quartets (module docstring).

    engine   a QuartetEngine with the replicate resident (device path), or None with `compute` given
    compute  injected per-rank compute `(quartets_local, subsample) -> (rstat, rscor, flags)` on host arrays
             (CPU tests of the partition / gather / regroup logic)
    dst      rank that receives the host arrays (None: every rank)
    nbuf     independent buffer sets: `start(b=k)` / `finish(b=k)` of different sets may be in flight together
    gather   "collective" (default): one all-gather per piece (RCCL over xGMI / gloo), every rank ends up with all rows on
             its gather device, the destination copies them D2H over ITS PCIe link.
             "host": no collective on the data path -- the destination's arrays live in ONE page-locked shared-memory
             segment that every rank of the node maps and registers with HIP; every rank copies its own rows D2H straight
             to their global positions over its OWN PCIe link (three copies per piece, no regrouping: a rank's part of a
             piece is a contiguous row range) and a barrier per batch tells the destination that all parts have landed.
             Built as the alternative to measure first on a multi-GPU node (on an 8-GPU step the
             collective path funnels 264 MB through rank 0's link); one node only; `device_rows()` is not available; the
             returned arrays are views of the segment, valid until the next start() on the same buffer set.
    """

    def __init__(self, Q: int, *, engine=None, compute: Optional[Callable] = None, group=None, device=None,
                 pieces: Optional[int] = None, dst: Optional[int] = None, nbuf: int = 1, collective_always: bool = False,
                 gather: str = "collective"):
        import torch
        import torch.distributed as dist
        if (engine is None) == (compute is None):
            raise ValueError("give either `engine` (device path) or `compute` (injected host compute)")
        if gather not in ("collective", "host"):
            raise ValueError("gather must be 'collective' or 'host'")
        self.gather = gather
        self.torch, self.dist, self.group = torch, dist, group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self.backend = dist.get_backend(group)
        else:
            self.world, self.rank, self.backend = 1, 0, None
        self.engine, self.compute, self.dst = engine, compute, dst
        # a process group of ONE rank normally skips the collective; `collective_always` issues it anyway (a
        # one-GPU rehearsal of the RCCL code path: communicator set-up, uint8 all-gather, stream hand-over)
        self.collective = (self.world > 1 or (collective_always and self.backend is not None)) and gather == "collective"
        self.plan = ShardPlan(Q, self.world, pieces)
        self.Q = int(Q)
        self.ranges = self.plan.local_ranges(self.rank)
        self.offsets = np.concatenate([[0], np.cumsum([hi - lo for lo, hi in self.ranges])]).astype(np.int64)
        self.n_local = int(self.offsets[-1])
        if engine is not None:
            idx = engine.device_id if device is None else torch.device(device).index
            self.dev = torch.device("cuda", idx)
        else:
            self.dev = torch.device("cpu")
        # where the collective runs: on the GPUs with RCCL, on the host with gloo
        self.gdev = self.dev if (self.backend == "nccl" or not self.collective) else torch.device("cpu")
        self.on_gpu = self.gdev.type == "cuda"
        P = self.plan
        self.sets = []
        for _ in range(nbuf):
            s = dict(
                slabs=[torch.zeros(P.slab_bytes(i), dtype=torch.uint8, device=self.dev) for i in range(P.npieces)],
                gathered=[torch.zeros(self.world * P.slab_bytes(i), dtype=torch.uint8, device=self.gdev)
                          if self.collective else None for i in range(P.npieces)],
                # the whole batch in global order (every rank holds it after the gathers)
                all_rstat=torch.zeros(P.rows_padded * 8, dtype=torch.uint8, device=self.gdev),
                all_rscor=torch.zeros(P.rows_padded * 24, dtype=torch.uint8, device=self.gdev),
                all_flags=torch.zeros(P.rows_padded, dtype=torch.uint8, device=self.gdev),
                host=None, done=None, works=[])
            self.sets.append(s)
        self.side = torch.cuda.Stream(self.dev) if self.on_gpu else None
        self.d_q = None
        self.q_local = None
        self._shm = []
        if gather == "host":
            self._open_shared_host(nbuf)

    # -- gather="host": the destination arrays in a shared, page-locked segment -----------------------
    def _open_shared_host(self, nbuf: int):
        """One POSIX shared-memory segment per buffer set: [rstat 8 B | rscor 24 B | flags 1 B] x rows_padded.  The
        creator is the destination rank (rank 0 when every rank wants the rows); its name goes to the others through
        the process group; every rank with a GPU registers the mapping with HIP so that its D2H copies are DMA."""
        from multiprocessing import shared_memory
        torch, dist, P = self.torch, self.dist, self.plan
        owner = self.dst if self.dst is not None else 0
        rows = P.rows_padded
        self._host_views = []
        for b in range(nbuf):
            nbytes = rows * 33 + 64
            names = [None]
            if self.rank == owner:
                # a segment larger than what /dev/shm can hold is created without complaint and kills the first process
                # that writes past the limit (SIGBUS): check before creating, and tell every rank the same thing
                try:
                    st = os.statvfs("/dev/shm")
                    room = st.f_bavail * st.f_frsize
                except OSError:
                    room = None
                if room is not None and room < nbytes + (16 << 20):
                    names[0] = f"!/dev/shm has {room >> 20} MiB free, the result segment needs {nbytes >> 20} MiB"
                else:
                    seg = shared_memory.SharedMemory(create=True, size=nbytes)
                    names[0] = seg.name
            if self.world > 1:
                dist.broadcast_object_list(names, src=owner, group=self.group)
            if names[0] is None or names[0].startswith("!"):
                self.close()
                raise RuntimeError("gather='host': " + (names[0] or "no segment")[1:])
            if self.rank != owner:
                seg = shared_memory.SharedMemory(name=names[0])
                try:        # attaching registers the segment with this process's resource tracker, which would unlink it
                    from multiprocessing import resource_tracker      # (again) at exit: only the owner unlinks
                    resource_tracker.unregister(seg._name, "shared_memory")
                except Exception:
                    pass
            buf = np.frombuffer(seg.buf, dtype=np.uint8, count=rows * 33)
            registered = False
            if self.on_gpu or self.engine is not None:
                rc = torch.cuda.cudart().cudaHostRegister(buf.ctypes.data, buf.nbytes, 0)
                registered = int(rc) == 0
            self._shm.append((seg, buf.ctypes.data, registered, self.rank == owner))
            self._host_views.append((buf[:rows * 8], buf[rows * 8:rows * 32], buf[rows * 32:rows * 33]))
        if self.world > 1:
            dist.barrier(group=self.group)           # every rank has attached before the owner may ever unlink

    def close(self):
        """Release the shared host segments of gather="host" (no-op otherwise)."""
        shm, self._shm = self._shm, []
        self._host_views = []
        for seg, ptr, registered, owner in shm:
            try:
                if registered:
                    self.torch.cuda.synchronize()
                    self.torch.cuda.cudart().cudaHostUnregister(ptr)
            except Exception:
                pass
            try:
                seg.close()
                if owner:
                    seg.unlink()
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- input ---------------------------------------------------------------------------------
    def set_quartets(self, quartets: np.ndarray):
        """`quartets` u32[Q,4], identical on every rank; the rank keeps (and uploads) its own rows."""
        quartets = np.asarray(quartets).reshape(-1, 4)
        if quartets.shape[0] != self.Q:
            raise ValueError(f"resolver was built for {self.Q} quartets, got {quartets.shape[0]}")
        parts = [quartets[lo:hi] for lo, hi in self.ranges]
        self.q_local = np.ascontiguousarray(np.concatenate(parts, axis=0), dtype=np.uint32)
        if self.engine is not None:
            self.d_q = self.torch.from_numpy(self.q_local.view(np.int32)).to(self.dev)

    def set_ranks(self, ranks: np.ndarray):
        """Lexicographic ranks (combinations.py:94-114) of the Q sampled quartets, identical on every rank:
        the rank uploads its own 8 bytes per quartet and unranks them on the device."""
        if self.engine is None:
            raise ValueError("set_ranks needs the device path")
        torch = self.torch
        ranks = np.asarray(ranks)
        if ranks.shape[0] != self.Q:
            raise ValueError(f"resolver was built for {self.Q} quartets, got {ranks.shape[0]}")
        local = np.ascontiguousarray(np.concatenate([ranks[lo:hi] for lo, hi in self.ranges]), dtype=np.int64)
        d_r = torch.from_numpy(local).to(self.dev)
        self.d_q = torch.empty((max(1, self.n_local), 4), dtype=torch.int32, device=self.dev)
        if self.n_local:
            self.engine.unrank_dev(d_r.data_ptr(), self.n_local, self.d_q.data_ptr(),
                                   torch.cuda.current_stream(self.dev).cuda_stream)
        self.q_local = None

    def quartet_buffer(self):
        """The rank's device quartet buffer int32[n_local,4] (allocated on first use), for producers that fill
        it on the device themselves (the replicate loop: unranking or the device sampler)."""
        if self.engine is None:
            raise ValueError("quartet_buffer needs the device path")
        if self.d_q is None or self.d_q.shape[0] < max(1, self.n_local):
            self.d_q = self.torch.empty((max(1, self.n_local), 4), dtype=self.torch.int32, device=self.dev)
        return self.d_q

    # -- one batch -------------------------------------------------------------------------------
    def _wants_host(self) -> bool:
        return self.dst is None or self.dst == self.rank

    def start(self, subsample_snps: bool = True, b: int = 0):
        """Enqueue one pass over the batch (kernels, gathers, regrouping, D2H); returns without waiting."""
        torch, dist, P = self.torch, self.dist, self.plan
        s = self.sets[b]
        if s["done"] is not None or s["works"]:
            raise RuntimeError("buffer set still in flight: call finish() first")
        want_host = self._wants_host()
        if self.gather == "host":
            if self.world > 1:
                dist.barrier(group=self.group)     # the destination is done with the rows of the previous batch in this segment
            s["host"] = self._host_views[b] if want_host else None
            self._start_host_gather(s, self._host_views[b], subsample_snps)
            return
        if want_host:
            if self.on_gpu:
                from .engine import pinned_empty
                host = (pinned_empty(P.rows_padded * 8, np.uint8), pinned_empty(P.rows_padded * 24, np.uint8),
                        pinned_empty(P.rows_padded, np.uint8))
            else:
                host = (s["all_rstat"].numpy(), s["all_rscor"].numpy(), s["all_flags"].numpy())
            s["host"] = host
            h_t = [torch.from_numpy(h) for h in host] if self.on_gpu else None
        if self.engine is not None:
            cur = torch.cuda.current_stream(self.dev)
            if self.n_local:
                self.engine.scan_dev(self.d_q.data_ptr(), self.n_local, subsample_snps, cur.cuda_stream)
        else:
            rstat, rscor, flags = self.compute(self.q_local, subsample_snps) if self.n_local else (
                np.zeros((0, 2), np.uint32), np.zeros((0, 3), np.float64), np.zeros(0, np.uint8))
        for i in range(P.npieces):
            p, blk = P.part[i], P.slab_bytes(i)
            lo, hi = self.ranges[i]
            n, off = hi - lo, int(self.offsets[i])
            slab = s["slabs"][i]
            if self.engine is not None:
                if n:
                    base = slab.data_ptr()
                    self.engine.svd_dev(off, n, base, base + 8 * p, base + 32 * p, cur.cuda_stream)
            else:
                sl = slab.numpy()
                sl[:8 * n] = np.ascontiguousarray(rstat[off:off + n], dtype=np.uint32).view(np.uint8).ravel()
                sl[8 * p:8 * p + 24 * n] = np.ascontiguousarray(rscor[off:off + n], dtype=np.float64).view(np.uint8).ravel()
                sl[32 * p:32 * p + n] = np.asarray(flags[off:off + n], dtype=np.uint8)
            # ONE collective per piece: every rank receives the parts of all ranks
            work = None
            if self.collective:
                src = slab if slab.device == self.gdev else slab.cpu()        # gloo rehearsal with the engine
                work = dist.all_gather_into_tensor(s["gathered"][i], src, group=self.group, async_op=True)
                g = s["gathered"][i].view(self.world, blk)
            else:
                g = slab.view(1, blk)
            rows = self.world * p
            r0 = P.start[i]

            def regroup_and_copy(g=g, p=p, rows=rows, r0=r0, work=work):
                if work is not None:
                    work.wait()                 # nccl: the current (side) stream waits; gloo: the host waits
                s["all_rstat"][8 * r0:8 * (r0 + rows)].view(self.world, 8 * p).copy_(g[:, :8 * p])
                s["all_rscor"][24 * r0:24 * (r0 + rows)].view(self.world, 24 * p).copy_(g[:, 8 * p:32 * p])
                s["all_flags"][r0:r0 + rows].view(self.world, p).copy_(g[:, 32 * p:33 * p])
                if want_host and self.on_gpu:
                    h_t[0][8 * r0:8 * (r0 + rows)].copy_(s["all_rstat"][8 * r0:8 * (r0 + rows)], non_blocking=True)
                    h_t[1][24 * r0:24 * (r0 + rows)].copy_(s["all_rscor"][24 * r0:24 * (r0 + rows)], non_blocking=True)
                    h_t[2][r0:r0 + rows].copy_(s["all_flags"][r0:r0 + rows], non_blocking=True)

            if self.on_gpu:
                if work is None:
                    self.side.wait_stream(cur)                 # the piece's kernels
                with torch.cuda.stream(self.side):
                    regroup_and_copy()
            else:
                if self.engine is not None and not self.collective:
                    torch.cuda.synchronize(self.dev)
                regroup_and_copy()
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(self.side)
            s["done"] = ev
        else:
            s["done"] = True

    def _start_host_gather(self, s, views, subsample_snps: bool):
        """gather="host": kernels of the rank's parts, then its rows straight to their global positions in the shared
        page-locked arrays (a part of a piece is the contiguous global row range part_range(i, rank))."""
        torch, P = self.torch, self.plan
        h_rstat, h_rscor, h_flags = (torch.from_numpy(v) for v in views)
        if self.engine is not None:
            cur = torch.cuda.current_stream(self.dev)
            if self.n_local:
                self.engine.scan_dev(self.d_q.data_ptr(), self.n_local, subsample_snps, cur.cuda_stream)
        else:
            rstat, rscor, flags = self.compute(self.q_local, subsample_snps) if self.n_local else (
                np.zeros((0, 2), np.uint32), np.zeros((0, 3), np.float64), np.zeros(0, np.uint8))
        for i in range(P.npieces):
            p = P.part[i]
            lo, hi = self.ranges[i]
            n, off = hi - lo, int(self.offsets[i])
            if not n:
                continue
            if self.engine is not None:
                slab = s["slabs"][i]
                base = slab.data_ptr()
                self.engine.svd_dev(off, n, base, base + 8 * p, base + 32 * p, cur.cuda_stream)
                self.side.wait_stream(cur)                     # the piece's kernels
                with torch.cuda.stream(self.side):
                    h_rstat[8 * lo:8 * hi].copy_(slab[:8 * n], non_blocking=True)
                    h_rscor[24 * lo:24 * hi].copy_(slab[8 * p:8 * p + 24 * n], non_blocking=True)
                    h_flags[lo:hi].copy_(slab[32 * p:32 * p + n], non_blocking=True)
            else:
                views[0][8 * lo:8 * hi] = np.ascontiguousarray(rstat[off:off + n], dtype=np.uint32).view(np.uint8).ravel()
                views[1][24 * lo:24 * hi] = np.ascontiguousarray(rscor[off:off + n], dtype=np.float64).view(np.uint8).ravel()
                views[2][lo:hi] = np.asarray(flags[off:off + n], dtype=np.uint8)
        if self.engine is not None:
            ev = torch.cuda.Event()
            ev.record(self.side)
            s["done"] = ev
        else:
            s["done"] = True

    def finish(self, b: int = 0):
        """Wait for set `b`; returns (rstat u32[Q,2], rscor f64[Q,3], flags u8[Q]) host arrays on the
        destination rank(s), (None, None, None) elsewhere."""
        s = self.sets[b]
        if s["done"] is None:
            raise RuntimeError("nothing in flight: call start() first")
        if self.gather == "host":
            if s["done"] is not True:
                s["done"].synchronize()                        # this rank's own copies have landed ...
            s["done"] = None
            if self.world > 1:
                self.dist.barrier(group=self.group)            # ... and so have everybody else's
            host, s["host"] = s["host"], None
            if host is None:
                return None, None, None
            Q = self.Q
            return (host[0][:8 * Q].view(np.uint32).reshape(Q, 2), host[1][:24 * Q].view(np.float64).reshape(Q, 3),
                    host[2][:Q])
        if self.on_gpu:
            s["done"].synchronize()
        s["done"] = None
        host, s["host"] = s["host"], None
        if host is None:
            return None, None, None
        Q = self.Q
        rstat = host[0][:8 * Q].view(np.uint32).reshape(Q, 2)
        rscor = host[1][:24 * Q].view(np.float64).reshape(Q, 3)
        flags = host[2][:Q]
        if not self.on_gpu:                                     # the all_* buffers are re-used by the next batch
            rstat, rscor, flags = rstat.copy(), rscor.copy(), flags.copy()
        return rstat, rscor, flags

    def resolve(self, subsample_snps: bool = True):
        self.start(subsample_snps)
        return self.finish()

    def local_rows(self, b: int = 0):
        """This rank's OWN rows as they left its kernels (valid after finish, either gather mode), in the order of
        plan.local_index(rank): (rstat int32[n,2], rscor f64[n,3], flags u8[n]) torch tensors on the compute device."""
        s, P, torch = self.sets[b], self.plan, self.torch
        a, c, f = [], [], []
        for i in range(P.npieces):
            lo, hi = self.ranges[i]
            n, p = hi - lo, P.part[i]
            slab = s["slabs"][i]
            a.append(slab[:8 * n].view(torch.int32).view(n, 2))
            c.append(slab[8 * p:8 * p + 24 * n].view(torch.float64).view(n, 3))
            f.append(slab[32 * p:32 * p + n])
        return torch.cat(a), torch.cat(c), torch.cat(f)

    def device_rows(self, b: int = 0):
        """The gathered batch in global order as it sits on this rank's gather device (valid after finish):
        (rstat int32[Q,2], rscor f64[Q,3], flags u8[Q]) torch views."""
        if self.gather == "host":
            raise RuntimeError("device_rows() needs gather='collective': with gather='host' no rank holds all rows on its device")
        s, Q, torch = self.sets[b], self.Q, self.torch
        return (s["all_rstat"][:8 * Q].view(torch.int32).view(Q, 2), s["all_rscor"][:24 * Q].view(torch.float64).view(Q, 3),
                s["all_flags"][:Q])


_resolvers: dict = {}


def resolve_sharded(tmparr, tmpmap, quartets, subsample_snps: bool = True, *, group=None,
                    compute: Optional[Callable] = None, device=None, dst: Optional[int] = None,
                    pieces: Optional[int] = None, gather: str = "collective"):
    """Resolve `quartets` (u32[Q,4], identical on every rank) across the ranks of `group`.

    Returns (quartets, rstat u32[Q,2], rscor f64[Q,3], flags u8[Q]), rows in the order of `quartets`,
    on every rank (`dst=None`) or on rank `dst` only (the other ranks get None for the three arrays: the
    reference's distributor consumes the rows on the client alone, run_inference.py:232-244).  With an
    uninitialised process group this is the 1-GPU path.  `compute(tmparr, tmpmap, quartets, subsample)`
    replaces the HIP engine in CPU tests of the control flow.  `gather="host"`: no collective, every rank copies its
    rows into a shared page-locked host array (ShardedResolver); the arrays returned are then COPIES of that segment."""
    import torch
    import torch.distributed as dist

    quartets = np.ascontiguousarray(quartets, dtype=np.uint32).reshape(-1, 4)
    Q = quartets.shape[0]
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1 and compute is None:
        dev_id = torch.cuda.current_device() if device is None else torch.device(device).index
        rstat, rscor, flags = gpu_compute(dev_id)(tmparr, tmpmap, quartets, subsample_snps)
        return quartets, rstat, rscor, flags
    if Q == 0:
        return quartets, np.zeros((0, 2), np.uint32), np.zeros((0, 3), np.float64), np.zeros(0, np.uint8)
    if compute is not None:
        res = ShardedResolver(Q, compute=lambda q, sub: compute(tmparr, tmpmap, q, sub), group=group, dst=dst,
                              pieces=pieces, gather=gather)
    else:
        from .resolve_quartets import _ensure_resident, get_engine
        dev_id = torch.cuda.current_device() if device is None else torch.device(device).index
        eng = get_engine(dev_id)
        _ensure_resident(eng, tmparr, tmpmap)
        key = (Q, world, dev_id, dst, pieces, gather)
        res = _resolvers.get("res") if _resolvers.get("key") == key else None
        # the cached resolver is only good for the SAME live engine and process group objects (an id() can be
        # re-used by a new group after the old one is destroyed; get_engine may hand out a fresh engine after close)
        if res is not None and not (res.engine is eng and res.group is group):
            res = None
        if res is None:                                  # buffers are kept for the next batch of this size
            old = _resolvers.get("res")
            if old is not None:
                old.close()
            res = ShardedResolver(Q, engine=eng, group=group, device=dev_id, dst=dst, pieces=pieces, gather=gather)
            _resolvers.update(key=key, res=res)
    res.set_quartets(quartets)
    rstat, rscor, flags = res.resolve(subsample_snps)
    if gather == "host":
        if rstat is not None:                            # views of the shared segment: hand out copies
            rstat, rscor, flags = rstat.copy(), rscor.copy(), flags.copy()
        if compute is not None:
            res.close()
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


def format_tsv_pieces(rqrts: np.ndarray, rscor: np.ndarray, rstat: np.ndarray, threads: int = 8, rows_per_piece: int = 1 << 18):
    """The same text as `format_tsv_bytes`, as a list of consecutive byte pieces formatted on a thread pool (the native
    formatter runs without the GIL): 5e6 rows take ~0.6 s on one core -- longer than the GPUs need to resolve them."""
    from concurrent.futures import ThreadPoolExecutor
    q = np.ascontiguousarray(rqrts, dtype=np.uint32).reshape(-1, 4)
    st = np.ascontiguousarray(rstat, dtype=np.uint32).reshape(-1, 2)
    sc = np.ascontiguousarray(rscor, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    if st.shape[0] != n or sc.shape[0] != n:
        raise ValueError("rqrts, rscor and rstat must have the same number of rows")
    if n <= rows_per_piece or threads <= 1:
        return [format_tsv_bytes(q, sc, st)]
    cuts = list(range(0, n, rows_per_piece)) + [n]
    with ThreadPoolExecutor(max_workers=int(threads)) as pool:
        return list(pool.map(lambda ab: format_tsv_bytes(q[ab[0]:ab[1]], sc[ab[0]:ab[1]], st[ab[0]:ab[1]]), zip(cuts[:-1], cuts[1:])))


def format_tsv(rqrts: np.ndarray, rscor: np.ndarray, rstat: np.ndarray) -> str:
    """Same text as the reference's
    ``pd.concat([DataFrame(rqrts), DataFrame(rscor), DataFrame(rstat)], axis=1)
      .to_csv(sep="\\t", float_format='%.6f', index=False, header=False)``
    (run_inference.py:233-234): 9 columns  a b c d score0 score1 score2 topo nsnps."""
    return format_tsv_bytes(rqrts, rscor, rstat).decode("ascii")


GATHER_ROWS = 1 << 23          # quartets resolved (and gathered) per collective round of `distributor`


def distributor(database_file: Path, qrts_file: Path, nsamples: int, qiter: Iterable, subsample_snps: bool,
                client=None, *, group=None, compute: Optional[Callable] = None,
                gather_rows: int = GATHER_ROWS) -> Path:
    """Mirror of run_inference.py:184-251: resolve every chunk of `qiter` and write the rows to
    `qrts_file`.  `client` (an ipyparallel Client in the reference) is accepted and ignored: the GPUs of
    the node take its place.  The reference hands chunk after chunk to whichever engine is free; here the
    chunks of one replicate are collected (up to `gather_rows` quartets: every replicate the reference's
    own size limits allow in practice) and resolved as ONE sharded batch -- one scan launch per rank, one
    all-gather per piece -- and only rank 0 receives the rows and writes the file."""
    import torch.distributed as dist
    from .resolve_quartets import load_database

    qrts_file = Path(qrts_file)
    rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
    if rank == 0:
        qrts_file.write_text("")                                   # :209
    tmparr, tmpmap = load_database(database_file)

    def flush(parts):
        qrts = parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)
        rqrts, rstat, rscor, _ = resolve_sharded(tmparr, tmpmap, qrts, subsample_snps, group=group,
                                                 compute=compute, dst=0)
        if rank == 0:
            with open(qrts_file, "ab") as out:                     # :240-244
                for piece in format_tsv_pieces(rqrts, rscor, rstat):
                    out.write(piece)

    parts, nrows = [], 0
    for chunk in qiter:                                            # :216-223
        qrts = np.array(list(chunk), dtype=np.uint32).reshape(-1, 4)
        if qrts.shape[0] == 0:
            continue
        parts.append(qrts)
        nrows += qrts.shape[0]
        if nrows >= gather_rows:
            flush(parts)
            parts, nrows = [], 0
    if parts:
        flush(parts)
    return qrts_file
