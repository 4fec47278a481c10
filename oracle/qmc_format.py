"""oracle/qmc_format.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Line-by-line restatement of the reference's wQMC input writer
(tetrad/src/run_inference.py:254-305 `iter_qmc_formatted`, :308-327 `write_qmc_format`): the
checker the native formatter `tq_format_qmc` (tetrad_amd/qmc_format.py, C ABI) is tested against.
Only ``tests/`` imports this module.

Same arithmetic on the same text: the reference re-reads the `%.6f`-rounded scores from the TSV,
so this restatement does too.  The reference shuffles the file with an unseeded `shuf` (:326-327);
here the shuffle is in-process and seedable.
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
