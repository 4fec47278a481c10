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

from .engine import QuartetEngine

_engines: dict[int, QuartetEngine] = {}
_resident: dict[int, tuple] = {}


def get_engine(device_id: int = 0) -> QuartetEngine:
    """Process-wide engine per device (the reference has one engine process per core)."""
    eng = _engines.get(device_id)
    if eng is None:
        eng = _engines[device_id] = QuartetEngine(device_id)
    return eng


def _ensure_resident(eng: QuartetEngine, tmparr: np.ndarray, tmpmap: np.ndarray):
    """Upload once per replicate, not once per chunk: skip the H2D when the same
    arrays (identity + shape) are already on the device."""
    key = (id(tmparr), id(tmpmap), tmparr.shape, tmpmap.shape)
    if _resident.get(eng.device_id, (None,))[0] != key:
        eng.set_data(tmparr, tmpmap)
        # keep the arrays alive so the ids stay unique while resident
        _resident[eng.device_id] = (key, tmparr, tmpmap)


def new_infer_resolved_quartets(tmparr, tmpmap, quartets, subsample_snps, device_id: int = 0):
    """resolve_quartets.py:191-265 -> (quartets u32[Q,4], rstat u32[Q,2], rscor f64[Q,3]).

    Zero-data quartets (no countable site): the reference stores an unseeded
    ``np.random.randint(3)`` as topology (:231); this engine stores 0.  Scores are
    0.001 in both (:232)."""
    eng = get_engine(device_id)
    tmparr = np.asarray(tmparr)
    tmpmap = np.asarray(tmpmap)
    _ensure_resident(eng, tmparr, tmpmap)
    quartets = np.ascontiguousarray(quartets, dtype=np.uint32).reshape(-1, 4)
    rstat, rscor, _ = eng.resolve(quartets, subsample_snps)
    return quartets, rstat, rscor


def infer_resolved_quartets(database: Path, nsamples: int, qrts, subsample_snps: bool = True,
                            device_id: int = 0):
    """resolve_quartets.py:17-39.  ``database`` is the project's HDF5 file with
    datasets ``tmparr`` and ``tmpmap`` (h5py), or an ``.npz`` with the same two arrays."""
    qrts = np.array(list(qrts), dtype=np.uint32)              # :28
    tmparr, tmpmap = load_database(database)                  # :33-35
    return new_infer_resolved_quartets(tmparr, tmpmap, qrts, subsample_snps, device_id)


_db_cache: dict = {}


def load_database(database):
    """Read tmparr/tmpmap; cached on (path, mtime) so chunk calls of one replicate
    reuse the arrays (and therefore the device-resident copy)."""
    path = Path(database)
    key = (str(path), path.stat().st_mtime_ns)
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
    return _db_cache["tmparr"], _db_cache["tmpmap"]


def _chunk_to_matrices(seqs, locus, mask, subsample: bool, device_id: int = 0):
    """Kernel-level interface: seqs u8[4,S], locus u32[S], mask[S] -> u32[3,16,16].

    The reference computes ``mask`` from ``seqs`` (:216-218) before calling the
    kernel; the device scan derives the same mask itself, so a caller-supplied
    mask is folded into the data by marking masked sites missing."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    if seqs.ndim != 2 or seqs.shape[0] != 4:
        raise ValueError("seqs must be u8[4,S]")
    seqs = seqs.copy()
    seqs[:, np.asarray(mask) != 0] = 78
    locus = np.ascontiguousarray(locus, dtype=np.uint32)
    with QuartetEngine(device_id) as eng:
        eng.set_data(seqs, locus)
        _, _, _, dbg = eng.resolve(np.array([[0, 1, 2, 3]], np.uint32), subsample, debug=True)
    return dbg["cmats"][0]


def subsample_chunk_to_matrices(seqs, locus, mask, device_id: int = 0):
    """resolve_quartets.py:42-73."""
    return _chunk_to_matrices(seqs, locus, mask, True, device_id)


def full_chunk_to_matrices(seqs, locus, mask, device_id: int = 0):
    """resolve_quartets.py:76-104."""
    return _chunk_to_matrices(seqs, locus, mask, False, device_id)
