#!/usr/bin/env python
"""Generate tests/golden/host_f2f3.npz by running the REFERENCE's own host functions either side of the
hot path.  Build container only (needs /root/reference):

    python tests/golden/make_golden_host.py

What is executed, unmodified:
  * /root/reference/tetrad/src/combinations.py, imported by file path as it stands (it needs nothing that is
    missing here): `_index_to_combination` (:94-106) and `random_combination_sample_via_index` (:109-114) in both
    regimes of NumPy's `Generator.choice(replace=False)` -- Floyd's algorithm (size <= population/50) and the
    tail shuffle (size > population/50) -- for 16 / 64 / 128 / 256 taxa;
  * `get_chunksize` (:73-96) and `iter_qmc_formatted` (:254-305) of /root/reference/tetrad/src/run_inference.py.
    That module imports loguru / pandas / toytree / h5py / tetrad at the top, so it cannot be imported whole;
    the two function definitions are taken out of its syntax tree (ast) and compiled as they stand, with
    `np`, `Path` and a do-nothing `logger` in their namespace (they use nothing else).

Inputs of the formatter: the quartets TSV of the reference's own c1 rows (tests/golden/c1_T16_S5000.npz, made by
make_golden.py from resolve_quartets.py), written with the pandas call the reference uses for that file
(run_inference.py:233-234).

What is stored: inputs and the reference's outputs.  Data only -- no reference source text.
"""
from __future__ import annotations

import ast
import importlib.util
import sys
import tempfile
from math import comb
from pathlib import Path

import numpy as np
import pandas as pd

REF_SRC = Path("/root/reference/tetrad/src")
OUT = Path(__file__).resolve().parent


def load_combinations():
    spec = importlib.util.spec_from_file_location("ref_combinations", REF_SRC / "combinations.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_run_inference_functions(names):
    tree = ast.parse((REF_SRC / "run_inference.py").read_text())
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names)

    class Quiet:
        def __getattr__(self, _):
            return lambda *a, **k: None

    ns = {"np": np, "Path": Path, "logger": Quiet()}
    exec(compile(ast.Module(body=picked, type_ignores=[]), str(REF_SRC / "run_inference.py"), "exec"), ns)
    return [ns[n] for n in names]


def reference_tsv(quartets, rscor, rstat) -> str:
    # run_inference.py:233-234 (the reference's writer of the file iter_qmc_formatted reads)
    return pd.concat([pd.DataFrame(quartets), pd.DataFrame(rscor), pd.DataFrame(rstat)], axis=1).to_csv(
        sep="\t", float_format="%.6f", index=False, header=False)


def sample_checksum(q: np.ndarray) -> np.ndarray:
    """Order-sensitive 64-bit checksum of u32[n,4] rows (taxa < 256): sum of packed rows x (position + 1)."""
    q = np.asarray(q, np.uint64)
    packed = (q[:, 0] << np.uint64(24)) | (q[:, 1] << np.uint64(16)) | (q[:, 2] << np.uint64(8)) | q[:, 3]
    with np.errstate(over="ignore"):
        return np.array((packed * (np.arange(len(q), dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64))


def main():
    C = load_combinations()
    get_chunksize, iter_qmc_formatted = load_run_inference_functions(["get_chunksize", "iter_qmc_formatted"])
    blob = dict(numpy_version=np.array(np.__version__))

    # ---- f2: unranking -------------------------------------------------------------------------------------
    for T in (16, 64, 128, 256):
        total = comb(T, 4)
        r = np.random.default_rng(T)
        idx = np.unique(np.concatenate([[0, 1, total - 1, total - 2, total // 2], r.integers(0, total, 300)]))
        blob[f"unrank_T{T}_ranks"] = idx.astype(np.uint64)
        blob[f"unrank_T{T}_quartets"] = np.array([C._index_to_combination(int(i), T) for i in idx], np.uint32)
    # ---- f2: the sampler, both regimes of Generator.choice -------------------------------------------------
    cases = [(16, 1000, 123), (16, 30, 5),            # tail shuffle (1000 > 1820/50) / Floyd (30 <= 36)
             (64, 30000, 7), (64, 5000, 8),           # tail / Floyd (5000 <= 12707)
             (128, 2000, 42), (128, 214000, 9),       # Floyd / tail (214000 > 213360)
             (256, 5000, 1)]                          # Floyd
    meta = []
    for T, size, seed in cases:
        q = np.array(C.random_combination_sample_via_index(T, size, seed), np.uint32)
        assert q.shape == (size, 4)
        regime = "tail" if size > comb(T, 4) // 50 else "floyd"
        meta.append((T, size, seed, regime == "tail"))
        if size > 50000:            # a large sample is stored as every 37th row + a position-weighted checksum of all rows
            blob[f"sample_T{T}_n{size}_s{seed}_every37"] = q[::37].astype(np.uint8)
            blob[f"sample_T{T}_n{size}_s{seed}_checksum"] = sample_checksum(q)
        else:
            blob[f"sample_T{T}_n{size}_s{seed}"] = q.astype(np.uint8)
        print(f"sample T={T} size={size} seed={seed}: {regime}")
    blob["sample_cases"] = np.array(meta, np.int64)
    # iter_chunks_random: chunk boundaries of the same draw
    chunks = list(C.iter_chunks_random(128, 2000, 300, 42))
    blob["chunks_T128_n2000_m300_s42_lens"] = np.array([len(c) for c in chunks], np.int64)
    # ---- distributor's chunk size ---------------------------------------------------------------------------
    nq = [1, 4, 999, 4999, 5000, 5001, 100000, 100001, 500000, 500001, 635376, 1000000, 5000000, 5000001, 174792640]
    nc = [1, 2, 4, 7, 8, 16, 40, 64]
    blob["chunksize_nquartets"] = np.array(nq, np.int64)
    blob["chunksize_ncores"] = np.array(nc, np.int64)
    blob["chunksize"] = np.array([[get_chunksize(a, b) for b in nc] for a in nq], np.int64)
    # ---- f3: wQMC lines from the reference's c1 rows --------------------------------------------------------
    g = np.load(OUT / "c1_T16_S5000.npz")
    settings = [(w, ms, mr) for w in (0, 1, 2, 3) for ms, mr in ((0, 1.0), (850, 1.0), (0, 5.0), (1800, 3.0))]
    blob["qmc_settings"] = np.array(settings, np.float64)
    with tempfile.TemporaryDirectory() as td:
        for mode in ("sub", "full"):
            f = Path(td) / f"{mode}.tsv"
            text = reference_tsv(g["quartets"], g[f"{mode}_rscor"], g[f"{mode}_rstat"])
            f.write_text(text)
            blob[f"qmc_{mode}_tsv"] = np.frombuffer(text.encode("ascii"), np.uint8)
            for k, (w, ms, mr) in enumerate(settings):
                lines = list(iter_qmc_formatted(f, w, ms, mr))
                blob[f"qmc_{mode}_lines_{k}"] = np.frombuffer(("\n".join(lines) + "\n").encode("ascii"), np.uint8)
                print(f"qmc {mode} weights={w} min_snps={ms} min_ratio={mr}: {len(lines)} lines")
    np.savez_compressed(OUT / "host_f2f3.npz", **blob)
    print("wrote host_f2f3.npz", (OUT / "host_f2f3.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
