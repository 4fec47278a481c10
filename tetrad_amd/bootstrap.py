"""Bootstrap replicates: host-side mirror of tetrad/src/run_inference.py:99-143.

The reference rewrites the `tmparr` / `tmpmap` datasets of its HDF5 database for every replicate
(resample loci with replacement, shuffle the columns of each locus, resolve IUPAC codes at random,
recode ACGT -> 0..3).  Here the project's `seqarr` and `spans` are uploaded once
(`QuartetEngine.set_source`) and each replicate is built on the GPU (`tq_bootstrap`): nothing is
written back to the database and nothing crosses PCIe but the nloci resampled locus indices.

The draws on the PROJECT's NumPy Generator are kept in the reference's order -- locus indices
(:117), then the two integer seeds (:120, :123) -- so the Generator state saved in the project JSON
after a replicate (:405-407) advances exactly as in the reference, and the quartet sample drawn next
(combinations.py:113) is the same one.  The streams behind the two seeds are the engine's own.
"""
from __future__ import annotations

import numpy as np

from .engine import QuartetEngine


def get_spans(maparr: np.ndarray) -> np.ndarray:
    """[start,end) site ranges of the loci of a snpsmap whose locus column is made of contiguous
    runs -- what tetrad/jit/get_spans.py:11-48 returns for such a map."""
    loc = np.asarray(maparr)[:, 0]
    starts = np.flatnonzero(np.concatenate([[True], loc[1:] != loc[:-1]]))
    ends = np.concatenate([starts[1:], [loc.shape[0]]])
    return np.stack([starts, ends], axis=1).astype(np.int64)


def draw_replicate(nloci: int, rng):
    """The three draws of run_inference.py:117-123 on the project Generator, without touching the
    GPU: `(lidxs, seed_shuffle, seed_ambig)` for `engine.bootstrap`.  Lets a driver prepare replicate
    k+1 on the host while the kernels of replicate k run (replicate-sequential RNG
    order, overlapped host work)."""
    lidxs = rng.choice(nloci, nloci, replace=True)                 # :117
    seed_shuffle = int(rng.integers(2**31))                        # :120
    seed_ambig = int(rng.integers(2**31))                          # :123
    return lidxs, seed_shuffle, seed_ambig


def resample_tmp_database(engine: QuartetEngine, rng) -> int:
    """run_inference.py:99-143 on the device.  `engine.set_source(seqarr, spans)` must have been
    called.  Returns the replicate's number of sites; the replicate is resident on the GPU
    (use `engine.get_data()` to inspect it or to write it to a database)."""
    rng = np.random.default_rng(rng)                               # :105 (pass-through for a Generator)
    return engine.bootstrap(*draw_replicate(engine.nloci, rng))
