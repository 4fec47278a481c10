"""QuartetEngine: one HIP context on one GPU (thin object wrapper over the C ABI).

Holds one replicate's genotype matrix resident in HBM (`set_data`, once per
replicate) and resolves quartet chunks against it.  Mirrors the state an
ipyparallel engine has in the reference while it runs
`infer_resolved_quartets` (tetrad/src/resolve_quartets.py:17-39) -- minus the
per-chunk HDF5 re-read.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import TetradHipError


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class _PinnedBlock:
    """One block of the library's page-locked host pool (tq_host_alloc); returned to the pool when the
    last NumPy view of it is garbage-collected."""
    __slots__ = ("ptr", "nbytes", "_free", "__weakref__")

    def __init__(self, nbytes: int):
        lib = _lib.load()
        p = ctypes.c_void_p()
        rc = lib.tq_host_alloc(int(nbytes), ctypes.byref(p))
        if rc != 0 or not p.value:
            raise TetradHipError(rc or -6, f"tq_host_alloc({nbytes}) failed")
        self.ptr, self.nbytes, self._free = p.value, int(nbytes), lib.tq_host_free

    @property
    def __array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            self._free(ctypes.c_void_p(self.ptr))
        except Exception:  # interpreter shutdown
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """An uninitialised NumPy array in page-locked host memory from the library's pool: the copy engine
    reads / writes it directly, asynchronously under the kernels (no staging, no extra host pass)."""
    dtype = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    raw = np.asarray(_PinnedBlock(max(n, 1)))           # .base keeps the block alive
    return raw[:n].view(dtype).reshape(shape)


class QuartetEngine:
    def __init__(self, device_id: int = 0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.tq_create(ctypes.byref(h), int(device_id))
        if rc != 0:
            raise TetradHipError(rc, self._lib.tq_last_error(None).decode())
        self._h = h
        self.device_id = int(device_id)
        self.T = self.S = 0
        #: bumped whenever the resident replicate is replaced (set_data / bootstrap); lets the API
        #: mirrors notice that what they uploaded is no longer what the device holds
        self.data_generation = 0

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.tq_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise TetradHipError(rc, self._lib.tq_last_error(self._h).decode())

    # -- data --------------------------------------------------------------
    def set_data(self, tmparr: np.ndarray, tmpmap: np.ndarray):
        """tmparr u8[T,S]; tmpmap u32[S,2] (column 0 used) or a 1-D locus column."""
        tmparr = np.ascontiguousarray(tmparr, dtype=np.uint8)
        if tmparr.ndim != 2:
            raise ValueError("tmparr must be 2-D [ntaxa, nsites]")
        tmpmap = np.asarray(tmpmap)
        if tmpmap.ndim == 2:
            tm = np.ascontiguousarray(tmpmap, dtype=np.uint32)
            stride = tm.shape[1]
        else:
            tm = np.ascontiguousarray(tmpmap, dtype=np.uint32)
            stride = 1
        if tm.shape[0] != tmparr.shape[1]:
            raise ValueError("tmpmap rows must equal tmparr columns")
        T, S = tmparr.shape
        self.data_generation += 1
        self._check(self._lib.tq_set_data(self._h, _ptr(tmparr), T, S, _ptr(tm), stride))
        self.T, self.S = T, S

    # -- bootstrap replicates on the device -------------------------------------
    def set_source(self, seqarr: np.ndarray, spans: np.ndarray):
        """seqarr u8[T,S0] (ASCII, IUPAC codes allowed); spans i64[nloci,2] ([start,end) per locus)."""
        seqarr = np.ascontiguousarray(seqarr, dtype=np.uint8)
        spans = np.ascontiguousarray(spans, dtype=np.int64).reshape(-1, 2)
        T, S0 = seqarr.shape
        self._check(self._lib.tq_set_source(self._h, _ptr(seqarr), T, S0, _ptr(spans), spans.shape[0]))
        self.nloci = spans.shape[0]

    def bootstrap(self, lidxs: np.ndarray, seed_shuffle: int, seed_ambig: int, stream: int | None = None) -> int:
        """Build the replicate for resampled loci `lidxs` on the device; returns its number of sites.
        With `stream` (a hipStream_t address, 0 = default stream) the call only enqueues."""
        lidxs = np.ascontiguousarray(lidxs, dtype=np.int64)
        S = ctypes.c_int64()
        self.data_generation += 1
        if stream is None:
            self._check(self._lib.tq_bootstrap(self._h, _ptr(lidxs), lidxs.shape[0], int(seed_shuffle),
                                               int(seed_ambig), ctypes.byref(S)))
        else:
            self._check(self._lib.tq_bootstrap_async(self._h, _ptr(lidxs), lidxs.shape[0], int(seed_shuffle),
                                                     int(seed_ambig), ctypes.byref(S), stream or None))
        T = ctypes.c_int64()
        self._check(self._lib.tq_data_shape(self._h, ctypes.byref(T), None))
        self.T, self.S = T.value, S.value
        return S.value

    def get_data(self):
        """Resident replicate in the reference layout: (tmparr u8[T,S] 0..3/78, tmpmap u32[S,2])."""
        T, S = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.tq_data_shape(self._h, ctypes.byref(T), ctypes.byref(S)))
        tmparr = np.zeros((T.value, S.value), np.uint8)
        tmpmap = np.zeros((S.value, 2), np.uint32)
        self._check(self._lib.tq_get_data(self._h, _ptr(tmparr), _ptr(tmpmap)))
        return tmparr, tmpmap

    def set_option(self, name: str, value: int) -> int:
        self._check(self._lib.tq_set_option(self._h, name.encode(), int(value)))
        return int(value)

    # -- host-buffer API -----------------------------------------------------
    def resolve(self, quartets: np.ndarray, subsample_snps: bool = True, debug: bool = False):
        """Returns (rstat u32[Q,2], rscor f64[Q,3], flags u8[Q]) and, with ``debug``,
        a dict with cmats u32[Q,3,16,16], svds f64[Q,3,16], ranks i32[Q,3].
        The result arrays live in page-locked memory of the library's pool (ordinary NumPy arrays to the
        caller), so the result D2H runs under the kernels."""
        q = quartets
        if not (isinstance(q, np.ndarray) and q.dtype == np.uint32 and q.flags.c_contiguous):
            src = np.asarray(quartets)
            q = pinned_empty((src.size // 4, 4), np.uint32)      # the conversion copy lands in pinned memory
            q[...] = src.reshape(-1, 4)
        q = q.reshape(-1, 4)
        Q = q.shape[0]
        rstat = pinned_empty((Q, 2), np.uint32)
        rscor = pinned_empty((Q, 3), np.float64)
        flags = pinned_empty(Q, np.uint8)
        if debug:
            cm = np.zeros((Q, 3, 16, 16), np.uint32)
            sv = np.zeros((Q, 3, 16), np.float64)
            rk = np.zeros((Q, 3), np.int32)
            self._check(self._lib.tq_resolve_debug(
                self._h, _ptr(q), Q, int(bool(subsample_snps)), _ptr(rstat), _ptr(rscor),
                _ptr(flags), _ptr(cm), _ptr(sv), _ptr(rk)))
            return rstat, rscor, flags, dict(cmats=cm, svds=sv, ranks=rk)
        self._check(self._lib.tq_resolve(
            self._h, _ptr(q), Q, int(bool(subsample_snps)), _ptr(rstat), _ptr(rscor), _ptr(flags)))
        return rstat, rscor, flags

    def resolve_to_host(self, d_quartets: int, Q: int, subsample_snps: bool = True, out=None):
        """Quartets on the device (address), results to host arrays: (rstat, rscor, flags).  ``out`` may
        hold the three arrays to fill (any NumPy arrays of the right shape; pinned ones are written by the
        copy engine directly)."""
        if out is None:
            out = (pinned_empty((Q, 2), np.uint32), pinned_empty((Q, 3), np.float64), pinned_empty(Q, np.uint8))
        rstat, rscor, flags = out
        self._check(self._lib.tq_resolve_to_host(
            self._h, d_quartets, Q, int(bool(subsample_snps)), _ptr(rstat), _ptr(rscor), _ptr(flags)))
        return rstat, rscor, flags

    # -- device-pointer API (addresses as ints, e.g. torch.Tensor.data_ptr()) -------
    def resolve_dev(self, d_quartets: int, Q: int, subsample_snps: bool, d_rstat: int,
                    d_rscor: int, d_flags: int = 0, stream: int = 0):
        self._check(self._lib.tq_resolve_dev(
            self._h, d_quartets, Q, int(bool(subsample_snps)), d_rstat, d_rscor,
            d_flags or None, stream or None))

    def resolve_range_dev(self, first_rank: int, Q: int, subsample_snps: bool, d_quartets: int,
                          d_rstat: int, d_rscor: int, d_flags: int = 0, stream: int = 0):
        self._check(self._lib.tq_resolve_range_dev(
            self._h, first_rank, Q, int(bool(subsample_snps)), d_quartets or None, d_rstat,
            d_rscor, d_flags or None, stream or None))

    def scan_dev(self, d_quartets: int, Q: int, subsample_snps: bool, stream: int = 0):
        """Stage 1 (ordering + site scan) of quartets [0,Q) into the context's count slab."""
        self._check(self._lib.tq_scan_dev(self._h, d_quartets, Q, int(bool(subsample_snps)), stream or None))

    def svd_dev(self, q0: int, n: int, d_rstat: int, d_rscor: int, d_flags: int = 0, stream: int = 0):
        """Stage 2 for rows [q0, q0+n) of the scanned batch; the pointers are the outputs OF ROW q0."""
        self._check(self._lib.tq_svd_dev(self._h, q0, n, d_rstat, d_rscor, d_flags or None, stream or None))

    def sample_quartets_dev(self, seed: int, Q: int, d_quartets: int, d_ranks: int = 0, stream: int = 0):
        """Q distinct quartets drawn uniformly from C(T,4) on the device (opt-in sampler, not the project Generator's stream)."""
        self._check(self._lib.tq_sample_quartets_dev(self._h, int(seed) & (2**64 - 1), Q, d_ranks or None, d_quartets,
                                                     stream or None))

    def unrank_dev(self, d_ranks: int, Q: int, d_quartets: int, stream: int = 0):
        self._check(self._lib.tq_unrank_dev(self._h, d_ranks, Q, d_quartets, stream or None))

    # -- measurement -----------------------------------------------------------
    def timing_enable(self, on: bool = True):
        self._check(self._lib.tq_timing_enable(self._h, int(on)))

    def timing_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        self._check(self._lib.tq_timing_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def timing_read_split(self):
        """(total_ms, scan_ms, svd_ms, resolve_calls) since the last read."""
        tot, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_int64()
        self._check(self._lib.tq_timing_read_split(
            self._h, ctypes.byref(tot), ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return tot.value, a.value, b.value, n.value

    KERNEL_TAGS = ("order", "scan", "bidiag", "bdsqr", "score")

    def timing_read_kernels(self):
        """({"order","scan","bidiag","bdsqr","score"} -> summed ms, resolve_calls) since the last read."""
        ms = (ctypes.c_double * 5)()
        n = ctypes.c_int64()
        self._check(self._lib.tq_timing_read_kernels(self._h, ms, 5, ctypes.byref(n)))
        return dict(zip(self.KERNEL_TAGS, list(ms))), n.value

    def debug_fetch(self, which: str, n: int) -> np.ndarray:
        """Scratch of the last resolve call (test hook): 'cm' u32[n,256], 'de' f64[3n,32], 'sv' f64[3n,16]."""
        shape, dt, code = {"cm": ((n, 256), np.uint32, 0), "de": ((3 * n, 32), np.float64, 1),
                           "sv": ((3 * n, 16), np.float64, 2), "bdsqr_stats": ((8,), np.uint64, 3)}[which]
        out = np.zeros(shape, dt)
        self._check(self._lib.tq_debug_fetch(self._h, code, _ptr(out), out.nbytes))
        return out

    def debug_bdsqr(self, de: np.ndarray, reps: int = 5):
        """Test hook: the bidiagonal-QR kernel alone on `de` f64[nmat,32] in the given order ->
        (sv f64[nmat,16], rotation steps u32[nmat], sweeps u32[nmat], ms per launch)."""
        de = np.ascontiguousarray(de, dtype=np.float64)
        nmat = de.shape[0]
        sv = np.zeros((nmat, 16), np.float64)
        work = np.zeros(nmat, np.uint32)
        ms = ctypes.c_double()
        self._check(self._lib.tq_debug_bdsqr(self._h, _ptr(de), nmat, _ptr(sv), _ptr(work), reps, ctypes.byref(ms)))
        return sv, work & 0xFFFF, work >> 16, ms.value

    def device_info(self):
        cu = ctypes.c_int32()
        w = ctypes.c_int32()
        pitch = ctypes.c_int64()
        self._check(self._lib.tq_device_info(self._h, ctypes.byref(cu), ctypes.byref(w), ctypes.byref(pitch)))
        return dict(num_cu=cu.value, waves_per_cu=w.value, row_pitch=pitch.value)
