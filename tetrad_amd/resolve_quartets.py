"""Host-side mirror of the reference worker interface, backed by the HIP engine.

Same names, argument meaning and return triple as
tetrad/src/resolve_quartets.py:

  infer_resolved_quartets(database, nsamples, qrts, subsample_snps)   (:17-39)
  new_infer_resolved_quartets(tmparr, tmpmap, quartets, subsample_snps) (:191-265)
  subsample_chunk_to_matrices / full_chunk_to_matrices(seqs, locus, mask) (:42-104)

so a `distributor`-style caller (run_inference.py:219-232) and the TSV writer
are unchanged.  Errors surface as Python exceptions (TetradHipError), which is
how the reference's failures surface too (run_inference.py:231-237).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from ._lib import FLAG_INVALID_DIAGNOSTIC, FLAG_NO_CONVERGENCE
from .engine import QuartetEngine

_engines: dict[int, QuartetEngine] = {}
_resident: dict[int, tuple] = {}


def get_engine(device_id: int = 0) -> QuartetEngine:
    """Process-wide engine per device (the reference has one engine process per core)."""
    eng = _engines.get(device_id)
    if eng is None:
        eng = _engines[device_id] = QuartetEngine(device_id)
    return eng


def _fingerprint(a: np.ndarray) -> tuple:
    """Content fingerprint of an array (xxh3-64 over the bytes; ~0.1 ms per MB)."""
    a = np.ascontiguousarray(a)
    try:
        import xxhash
        h = xxhash.xxh3_64_intdigest(memoryview(a).cast("B"))
    except ImportError:  # pragma: no cover
        import zlib
        h = zlib.crc32(memoryview(a).cast("B"))
    return (a.shape, a.dtype.str, h)


def _ensure_resident(eng: QuartetEngine, tmparr: np.ndarray, tmpmap: np.ndarray, replicate_token=None):
    """Upload once per replicate, not once per chunk.  The H2D is skipped only when the device still
    holds what this function uploaded last (the engine's data generation is unchanged: nobody called
    `set_data` / `bootstrap` on it in between) AND the arrays are the same replicate: equal
    `replicate_token` when the caller passes one (any hashable; e.g. the bootstrap index), otherwise
    equal content fingerprints -- so an array refilled in place between replicates is uploaded again,
    as the reference re-reads its database on every call (resolve_quartets.py:33-35)."""
    key = ("token", replicate_token, tmparr.shape, tmpmap.shape) if replicate_token is not None \
        else ("content", _fingerprint(tmparr), _fingerprint(tmpmap))
    if _resident.get(eng.device_id) != (key, eng.data_generation):
        eng.set_data(tmparr, tmpmap)
        _resident[eng.device_id] = (key, eng.data_generation)


def invalidate(device_id: int | None = None):
    """Forget what is resident (all devices when `device_id` is None): the next call uploads again."""
    if device_id is None:
        _resident.clear()
    else:
        _resident.pop(device_id, None)


def new_infer_resolved_quartets(tmparr, tmpmap, quartets, subsample_snps, device_id: int = 0, replicate_token=None):
    """resolve_quartets.py:191-265 -> (quartets u32[Q,4], rstat u32[Q,2], rscor f64[Q,3]).

    Zero-data quartets (no countable site): the reference stores an unseeded
    ``np.random.randint(3)`` as topology (:231); this engine stores 0.  Scores are
    0.001 in both (:232)."""
    eng = get_engine(device_id)
    tmparr = np.asarray(tmparr)
    tmpmap = np.asarray(tmpmap)
    _ensure_resident(eng, tmparr, tmpmap, replicate_token)
    quartets = np.ascontiguousarray(quartets, dtype=np.uint32).reshape(-1, 4)
    rstat, rscor, flags = eng.resolve(quartets, subsample_snps)
    if (flags & FLAG_NO_CONVERGENCE).any():                       # np.linalg.svd raises the same (:242)
        raise np.linalg.LinAlgError("SVD did not converge")
    if (flags & FLAG_INVALID_DIAGNOSTIC).any():
        raise RuntimeError("the engine is in a timing-diagnostic mode (scan_method 2..5 or phases 1 / 2): its rows are "
                           "not results; reset the option before resolving")
    return quartets, rstat, rscor


def infer_resolved_quartets(database: Path, nsamples: int, qrts, subsample_snps: bool = True,
                            device_id: int = 0):
    """resolve_quartets.py:17-39.  ``database`` is the project's HDF5 file with
    datasets ``tmparr`` and ``tmpmap`` (h5py), or an ``.npz`` with the same two arrays."""
    qrts = np.array(list(qrts), dtype=np.uint32).reshape(-1, 4)   # :28
    tmparr, tmpmap, token = load_database(database, with_token=True)   # :33-35
    return new_infer_resolved_quartets(tmparr, tmpmap, qrts, subsample_snps, device_id, replicate_token=token)


_db_cache: dict = {}


def load_database(database, with_token: bool = False):
    """Read tmparr/tmpmap; cached on (path, mtime, size) so chunk calls of one replicate reuse the
    arrays (and therefore the device-resident copy).  The cached arrays are private to this module, so
    the cache key doubles as the replicate token of `_ensure_resident`."""
    path = Path(database)
    st = path.stat()
    key = (str(path.resolve()), st.st_mtime_ns, st.st_size)
    hit = _db_cache.get("key") == key
    if not hit:
        if path.suffix == ".npz":
            with np.load(path, allow_pickle=False) as z:
                tmparr, tmpmap = z["tmparr"], z["tmpmap"]
        else:
            import h5py  # the reference's storage format; not installed in every image
            with h5py.File(path, "r", swmr=True) as io5:
                tmparr = io5["tmparr"][:]
                tmpmap = io5["tmpmap"][:]
        _db_cache.update(key=key, tmparr=tmparr, tmpmap=tmpmap)
    if with_token:
        return _db_cache["tmparr"], _db_cache["tmpmap"], key
    return _db_cache["tmparr"], _db_cache["tmpmap"]


def _chunk_to_matrices(seqs, locus, mask, subsample: bool, device_id: int = 0):
    """Kernel-level interface: seqs u8[4,S], locus u32[S], mask[S] -> u32[3,16,16].

    The caller's ``mask`` is honoured exactly as the reference's count kernels honour it
    (resolve_quartets.py:59-64, :89-95): every site it leaves open is a candidate -- invariant sites
    included (engine option ``count_invariant``: the worker's own mask, :216-218, never leaves one open,
    but a caller's may) -- and every site it closes is skipped, whatever the bases there.  One
    difference, on input the reference does not define: a site left open whose bases are not all in
    0..3 (a missing 78 would index outside the 16x16 matrix there) is skipped here."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    if seqs.ndim != 2 or seqs.shape[0] != 4:
        raise ValueError("seqs must be u8[4,S]")
    seqs = seqs.copy()
    seqs[:, np.asarray(mask) != 0] = 78
    locus = np.ascontiguousarray(locus, dtype=np.uint32)
    with QuartetEngine(device_id) as eng:
        eng.set_option("count_invariant", 1)
        eng.set_data(seqs, locus)
        _, _, _, dbg = eng.resolve(np.array([[0, 1, 2, 3]], np.uint32), subsample, debug=True)
    return dbg["cmats"][0]


def subsample_chunk_to_matrices(seqs, locus, mask, device_id: int = 0):
    """resolve_quartets.py:42-73."""
    return _chunk_to_matrices(seqs, locus, mask, True, device_id)


def full_chunk_to_matrices(seqs, locus, mask, device_id: int = 0):
    """resolve_quartets.py:76-104."""
    return _chunk_to_matrices(seqs, locus, mask, False, device_id)
