"""wQMC input lines from the quartets TSV -- host-side mirror of
tetrad/src/run_inference.py:254-327 (the consumer right after the hot path).

Same arithmetic on the same text: the reference re-reads the `%.6f`-rounded scores from the TSV,
so this mirror does too (weights computed from full-precision scores would differ in the 6th
decimal).  The reference shuffles the file with an unseeded `shuf` (:326-327); here the shuffle is
in-process and seedable.
"""
from __future__ import annotations

from itertools import islice
from pathlib import Path

import numpy as np


def iter_qmc_formatted(qrts_file: Path, weights: int, min_snps: int = 0, min_ratio: float = 1.0):
    """run_inference.py:254-305.  Yields "a,b|c,d:weight" strings."""
    min_snps = max(1, min_snps)                                   # :258
    with open(qrts_file, "r") as datain:
        for line in datain:
            values = line.split("\t")
            order = int(values[7])                                # :264-270: topology -> split
            if order == 1:
                qrts = values[0], values[2], values[1], values[3]
            elif order == 2:
                qrts = values[0], values[3], values[1], values[2]
            else:
                qrts = values[:4]
            scores = np.array(values[4:7], dtype=np.float64)
            nsnps = int(values[8])
            if nsnps < min_snps:                                  # :275
                continue
            if not weights:                                       # :280-297
                weight = 1.0
                ratio = 1.0
            else:
                scores = np.array(sorted(scores))
                smean = scores[1:].mean()
                smin = scores.min()
                ratio = 1. if not smin else smean / smin
                if weights == 1:
                    weight = smean
                elif weights == 2:
                    weight = ratio
                elif weights == 3:
                    weight = 1. - smin / scores.sum()
                else:
                    raise ValueError(f"no weight strategy {weights}")
            if ratio < min_ratio:                                 # :300
                continue
            yield "{},{}|{},{}:{:.5f}".format(*qrts, weight)     # :305


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


def write_qmc_format(qrts_file: Path, qmc_in_file: Path, weights: int = 0, min_snps: int = 0,
                     min_ratio: float = 1.0, seed=None) -> None:
    """run_inference.py:308-327: formatted lines in random order (seedable instead of `shuf`)."""
    lines = list(iter_qmc_formatted(qrts_file, weights, min_snps, min_ratio))
    np.random.default_rng(seed).shuffle(lines)
    with open(qmc_in_file, "w") as out:
        it = iter(lines)
        while True:
            chunk = "\n".join(islice(it, 50_000))                 # :311
            if not chunk:
                break
            out.write(chunk + "\n")
