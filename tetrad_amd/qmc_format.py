"""wQMC input lines for the consumer right after the hot path.

The reference writes the quartets TSV, re-reads it line by line and formats "a,b|c,d:weight"
(tetrad/src/run_inference.py:254-327).  Here the lines are produced by the native formatter of the
C ABI (`tq_format_qmc`: weights 0-3, `min_snps`, `min_ratio`; the scores are rounded to six decimals
first, exactly as the reference reads them back from the file) -- straight from the result arrays,
or from a TSV parsed back into arrays.  The reference shuffles the file with an unseeded `shuf`
(:326-327); here the shuffle is in-process and seedable.

The line-by-line restatement of the reference's formatter that this module is tested against lives
in oracle/qmc_format.py (test infrastructure).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

def qmc_lines(rqrts: np.ndarray, rscor: np.ndarray, rstat: np.ndarray, weights: int = 0, min_snps: int = 0,
              min_ratio: float = 1.0) -> list[bytes]:
    """The lines `iter_qmc_formatted` yields for the TSV these arrays would be written to, straight
    from the arrays (native formatter of the C ABI: `tq_format_qmc`; the scores are rounded to six
    decimals first, exactly as the reference reads them back from the file)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    if weights not in (0, 1, 2, 3):
        raise ValueError(f"no weight strategy {weights}")
    q = np.ascontiguousarray(rqrts, dtype=np.uint32).reshape(-1, 4)
    st = np.ascontiguousarray(rstat, dtype=np.uint32).reshape(-1, 2)
    sc = np.ascontiguousarray(rscor, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    cap = 64 * n + 4096
    written, nlines = ctypes.c_int64(), ctypes.c_int64()
    for _ in range(2):
        buf = np.empty(cap, dtype=np.uint8)
        rc = lib.tq_format_qmc(q.ctypes.data, st.ctypes.data, sc.ctypes.data, n, int(weights), int(min_snps),
                               float(min_ratio), buf.ctypes.data, cap, ctypes.byref(written), ctypes.byref(nlines))
        if rc == 0:
            return buf[:written.value].tobytes().split(b"\n")[:-1]
        if rc != -6:
            raise _lib.TetradHipError(rc, "tq_format_qmc")
        cap = written.value
    raise _lib.TetradHipError(rc, "tq_format_qmc: buffer sizing failed")


def write_qmc_from_arrays(rqrts, rscor, rstat, qmc_in_file: Path, weights: int = 0, min_snps: int = 0,
                          min_ratio: float = 1.0, seed=None) -> int:
    """`write_qmc_format` without the round trip through the TSV text; returns the number of lines."""
    lines = qmc_lines(rqrts, rscor, rstat, weights, min_snps, min_ratio)
    order = np.random.default_rng(seed).permutation(len(lines))
    with open(qmc_in_file, "wb") as out:
        out.write(b"".join(lines[i] + b"\n" for i in order))
    return len(lines)


def read_quartets_tsv(qrts_file: Path):
    """The 9-column quartets TSV (run_inference.py:233-234) back into `(rqrts u32[n,4], rscor f64[n,3],
    rstat u32[n,2])`."""
    import pandas as pd
    cols = list(range(9))
    dtypes = {0: np.uint32, 1: np.uint32, 2: np.uint32, 3: np.uint32, 4: np.float64, 5: np.float64, 6: np.float64,
              7: np.uint32, 8: np.uint32}
    try:
        tab = pd.read_csv(qrts_file, sep="\t", header=None, names=cols, dtype=dtypes)
    except pd.errors.EmptyDataError:
        tab = pd.DataFrame({c: np.zeros(0, dtypes[c]) for c in cols})
    rqrts = np.ascontiguousarray(tab[[0, 1, 2, 3]].to_numpy(np.uint32))
    rscor = np.ascontiguousarray(tab[[4, 5, 6]].to_numpy(np.float64))
    rstat = np.ascontiguousarray(tab[[7, 8]].to_numpy(np.uint32))
    return rqrts, rscor, rstat


def iter_qmc_formatted(qrts_file: Path, weights: int, min_snps: int = 0, min_ratio: float = 1.0):
    """Same lines, same order as run_inference.py:254-305 yields for `qrts_file` (native formatter)."""
    rqrts, rscor, rstat = read_quartets_tsv(qrts_file)
    for line in qmc_lines(rqrts, rscor, rstat, weights, min_snps, min_ratio):
        yield line.decode("ascii")


def write_qmc_format(qrts_file: Path, qmc_in_file: Path, weights: int = 0, min_snps: int = 0,
                     min_ratio: float = 1.0, seed=None) -> int:
    """run_inference.py:308-327 for a TSV on disk; returns the number of lines written."""
    rqrts, rscor, rstat = read_quartets_tsv(qrts_file)
    return write_qmc_from_arrays(rqrts, rscor, rstat, qmc_in_file, weights, min_snps, min_ratio, seed)
