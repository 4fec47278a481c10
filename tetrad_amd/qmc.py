"""Quartet supertree: the step right after the hot path (second half).

The reference writes the wQMC input file and shells out to a prebuilt binary
(tetrad/src/run_inference.py:146-166 `run_qmc`, :330-357 `infer_supertree`).  Here the tree comes from the
library's own weighted Quartet MaxCut (`tq_qmc_tree`, host C++, written from the published method -- the
reference's binary has no source and is never executed; parity with it is unpinned by construction).
Tip labels are the taxon numbers, as in the file the reference's `relabel_tree` (:169-181) post-processes.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np


def qmc_tree(splits: np.ndarray, weights=None, ntaxa: int | None = None, seed: int = 0) -> str:
    """Newick of the supertree of quartets `splits` u32[n,4] ("a,b|c,d" per row) with optional weights."""
    from . import _lib
    lib = _lib.load()
    sp = np.ascontiguousarray(splits, dtype=np.uint32).reshape(-1, 4)
    n = sp.shape[0]
    if ntaxa is None:
        ntaxa = int(sp.max()) + 1 if n else 1
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
    if w is not None and w.shape[0] != n:
        raise ValueError("weights must have one entry per quartet")
    cap = 16 * int(ntaxa) + 64
    written = ctypes.c_int64()
    for _ in range(2):
        buf = np.empty(cap, dtype=np.uint8)
        rc = lib.tq_qmc_tree(sp.ctypes.data, None if w is None else w.ctypes.data, n, int(ntaxa), int(seed) & (2**64 - 1),
                             buf.ctypes.data, cap, ctypes.byref(written))
        if rc == 0:
            return buf[:written.value].tobytes().decode("ascii")
        if rc != -6 or written.value <= cap:
            raise _lib.TetradHipError(rc, "tq_qmc_tree")
        cap = written.value
    raise _lib.TetradHipError(rc, "tq_qmc_tree: buffer sizing failed")


def qmc_splits(rqrts, rscor, rstat, weights: int = 0, min_snps: int = 0, min_ratio: float = 1.0):
    """The rows `qmc_format.qmc_lines` would write, as arrays: (splits u32[n,4], weights f64[n]) -- same filters, same
    weight strategies, the weight as its "%.5f" text reads back -- without producing or parsing text."""
    from . import _lib
    lib = _lib.load()
    if weights not in (0, 1, 2, 3):
        raise ValueError(f"no weight strategy {weights}")
    q = np.ascontiguousarray(rqrts, dtype=np.uint32).reshape(-1, 4)
    st = np.ascontiguousarray(rstat, dtype=np.uint32).reshape(-1, 2)
    sc = np.ascontiguousarray(rscor, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    sp = np.empty((n, 4), np.uint32)
    w = np.empty(n, np.float64)
    kept = ctypes.c_int64()
    rc = lib.tq_qmc_splits(q.ctypes.data, st.ctypes.data, sc.ctypes.data, n, int(weights), int(min_snps), float(min_ratio),
                           sp.ctypes.data, w.ctypes.data, ctypes.byref(kept))
    if rc != 0:
        raise _lib.TetradHipError(rc, "tq_qmc_splits")
    return sp[:kept.value], w[:kept.value]


def parse_qmc_lines(lines) -> tuple[np.ndarray, np.ndarray]:
    """"a,b|c,d:w" lines (run_inference.py:305) -> (splits u32[n,4], weights f64[n])."""
    sp, w = [], []
    for ln in lines:
        ln = ln.decode() if isinstance(ln, bytes) else ln
        ln = ln.strip()
        if not ln:
            continue
        left, _, weight = ln.partition(":")
        ab, cd = left.split("|")
        sp.append([int(x) for x in ab.split(",")] + [int(x) for x in cd.split(",")])
        w.append(float(weight) if weight else 1.0)
    return np.array(sp, dtype=np.uint32).reshape(-1, 4), np.array(w, dtype=np.float64)


def run_qmc(qmc_in_file: Path, qmc_out_file: Path, use_weights: bool, ntaxa: int | None = None, seed: int = 0) -> None:
    """run_inference.py:146-166 without the external binary: reads the wQMC input file, writes the newick."""
    with open(qmc_in_file) as f:
        splits, weights = parse_qmc_lines(f)
    nwk = qmc_tree(splits, weights if use_weights else None, ntaxa, seed)
    Path(qmc_out_file).write_text(nwk + "\n")


def infer_supertree_from_arrays(rqrts, rscor, rstat, ntaxa: int, weights: int = 0, min_snps: int = 0,
                                min_ratio: float = 1.0, seed: int = 0) -> str:
    """run_inference.py:330-357 straight from the result arrays of a replicate: the wQMC lines of
    `tq_format_qmc` (same filters and weight strategies as :254-305), then the tree."""
    splits, w = qmc_splits(rqrts, rscor, rstat, weights, min_snps, min_ratio)
    return qmc_tree(splits, w if weights else None, ntaxa, seed)


def relabel_tree(newick: str, samples) -> str:
    """The tree with the numeric tip labels replaced by sample names -- what run_inference.py:169-181 does with
    toytree.  `samples` maps the taxon number to its name (a dict, or a sequence indexed by taxon number)."""
    import re
    names = samples if hasattr(samples, "get") else dict(enumerate(samples))

    def sub(m):
        name = names.get(int(m.group(2)))
        if name is None:
            raise KeyError(f"no sample name for taxon {m.group(2)}")
        name = str(name)
        if re.search(r"[\s(),:;\[\]']", name):
            name = "'" + name.replace("'", "''") + "'"
        return m.group(1) + name

    return re.sub(r"([(,])(\d+)(?=[,):])", sub, newick)


def infer_supertree(qrts_file: Path, qmc_in_file: Path, qmc_out_file: Path, ntaxa: int, weights: int = 0, min_snps: int = 0,
                    min_ratio: float = 1.0, samples=None, seed: int = 0) -> str:
    """run_inference.py:330-357 on explicit paths instead of a Project: the quartets TSV -> shuffled wQMC input file
    (`qmc_format.write_qmc_format`) -> tree file (`run_qmc`) -> newick with sample names when `samples` is given
    (`relabel_tree`), numeric tips otherwise."""
    from .qmc_format import write_qmc_format
    write_qmc_format(qrts_file, qmc_in_file, weights, min_snps, min_ratio, seed=seed)      # :347
    run_qmc(qmc_in_file, qmc_out_file, bool(weights), ntaxa=ntaxa, seed=seed)              # :350
    nwk = Path(qmc_out_file).read_text().strip()
    return relabel_tree(nwk, samples) if samples is not None else nwk                      # :353
