#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own hot-path functions.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What is executed: /root/reference/tetrad/src/resolve_quartets.py, loaded by file
path, unmodified.  Its two top-level imports that are not installed here are
satisfied as SURVEY.md section 8c describes: ``numba.njit`` becomes the identity
decorator (the two jitted functions are pure integer counting loops, so running
them interpreted is exact), ``h5py`` is an empty module (only the outer HDF5
wrapper :33-35 uses it and it is not called).  ``np.uint32(-1)`` at :58 raises
under NumPy 2.x where numba wraps to 4294967295, so the loaded module sees a
NumPy proxy whose ``uint32`` wraps like numba's.  The SVD / rank arithmetic is
plain ``numpy.linalg`` in the reference as well.

What is stored: inputs (tmparr, tmpmap, quartets) and the reference's outputs
(rstat, rscor from new_infer_resolved_quartets; cmats from the two count
kernels called exactly as :221/:223 call them; singular values from the same
numpy call as :242).  Data only -- no reference source text.

Zero-data quartets: the reference's topology there is an unseeded
``np.random.randint(3)`` (:231); those rows are recorded in ``zero_data`` and
their topology is not part of the contract.
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from tetrad_amd import synth  # noqa: E402

REF = Path("/root/reference/tetrad/src/resolve_quartets.py")
OUT = Path(__file__).resolve().parent


def load_reference():
    numba = types.ModuleType("numba")
    numba.njit = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules.setdefault("numba", numba)
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    spec = importlib.util.spec_from_file_location("ref_resolve_quartets", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class WrapU32(np.uint32):           # still a valid dtype (-> uint32)
        def __new__(cls, v=0):
            return np.uint32(int(v) & 0xFFFFFFFF)

    class NPProxy:
        uint32 = WrapU32

        def __getattr__(self, name):
            return getattr(np, name)

    mod.np = NPProxy()
    return mod


def run_reference(ref, tmparr, tmpmap, quartets, subsample, with_cmats=True):
    np.random.seed(0)  # only affects the zero-data randint, which is not contractual
    q, rstat, rscor = ref.new_infer_resolved_quartets(tmparr, tmpmap, quartets, subsample)
    out = dict(rstat=rstat, rscor=rscor)
    if with_cmats:
        Q = quartets.shape[0]
        cmats = np.zeros((Q, 3, 16, 16), np.uint32)
        svds = np.zeros((Q, 3, 16), np.float64)
        for i in range(Q):
            seqs = tmparr[quartets[i], :]
            nmask0 = np.sum(seqs >= 78, axis=0)
            nmask1 = np.sum(seqs == seqs[0], axis=0) == 4
            fn = ref.subsample_chunk_to_matrices if subsample else ref.full_chunk_to_matrices
            cmats[i] = fn(seqs, tmpmap[:, 0], nmask0 + nmask1)
            if cmats[i, 0].sum():
                for t in range(3):
                    svds[i, t] = np.linalg.svd(cmats[i, t].astype(np.float64))[1]
        out["cmats"] = cmats
        out["svds"] = svds
    out["zero_data"] = (rstat[:, 1] == 0)
    return out


def save(name, tmparr, tmpmap, quartets, ref, with_cmats=True):
    blob = dict(tmparr=tmparr, tmpmap=tmpmap, quartets=quartets,
                numpy_version=np.array(np.__version__))
    for sub in (False, True):
        res = run_reference(ref, tmparr, tmpmap, quartets, sub, with_cmats)
        tag = "sub" if sub else "full"
        for k, v in res.items():
            blob[f"{tag}_{k}"] = v
    np.savez_compressed(OUT / f"{name}.npz", **blob)
    print(f"wrote {name}.npz  T={tmparr.shape[0]} S={tmparr.shape[1]} Q={quartets.shape[0]}")


def loci_map(S, rng, mean=4):
    lens = 1 + rng.poisson(mean, size=S)
    loc = np.repeat(np.arange(S, dtype=np.uint32), lens)[:S]
    tmap = np.zeros((S, 2), np.uint32)
    tmap[:, 0] = loc
    tmap[:, 1] = np.arange(S)
    return tmap


def make_slices(ref):
    # (10) slices of c2 / c3 / c4 shaped inputs are regenerated from the seed on the fly by
    #      the tests (inputs are deterministic); only the reference outputs for a sample of
    #      quartets are stored here (round 4: 128 / 96 / 48 quartets, 24 / 12 / 12 before).
    for cfg, nq in (("c2", 128), ("c3", 96), ("c4", 48)):
        T, S, _ = synth.CONFIGS[cfg]
        arr, tmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
        qrts = synth.random_quartets(T, nq, seed=777)
        blob = dict(quartets=qrts, config=np.array(cfg), numpy_version=np.array(np.__version__))
        for sub in (False, True):
            res = run_reference(ref, arr, tmap, qrts, sub, with_cmats=True)
            tag = "sub" if sub else "full"
            for k, v in res.items():
                blob[f"{tag}_{k}"] = v
        np.savez_compressed(OUT / f"{cfg}_slice.npz", **blob)
        print(f"wrote {cfg}_slice.npz  Q={nq}")


def make_rows(ref):
    # (12) the reference worker's ROWS (rstat, rscor; no count matrices) for thousands of quartets of the c2 / c3 / c4
    #      benchmark inputs: the bulk parity evidence at benchmark size straight from the reference
    for cfg, nq in (("c2", 3000), ("c3", 2000), ("c4", 1000)):
        T, S, _ = synth.CONFIGS[cfg]
        arr, tmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
        qrts = synth.random_quartets(T, nq, seed=4321)
        blob = dict(quartets=qrts, config=np.array(cfg), numpy_version=np.array(np.__version__))
        for sub in (False, True):
            res = run_reference(ref, arr, tmap, qrts, sub, with_cmats=False)
            tag = "sub" if sub else "full"
            for k, v in res.items():
                blob[f"{tag}_{k}"] = v
        np.savez_compressed(OUT / f"{cfg}_rows.npz", **blob)
        print(f"wrote {cfg}_rows.npz  Q={nq}")


def main(only=None):
    ref = load_reference()
    if only == "rows":
        make_rows(ref)
        return
    if only == "slices":
        make_slices(ref)
        return
    if only == "minrank_T14_S600":          # (11) alone: the other fixtures are unchanged (they reproduce bit for bit)
        arr, tmap = synth.simulate_tmparr(14, 600, seed=9, p=0.01, missing=0.3)
        save("minrank_T14_S600", arr, tmap, synth.all_quartets(14), ref)
        return

    # (1) dense uniform-random data, T=8, S=400 (S not a multiple of 64), all 70 quartets
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 4, size=(8, 400), dtype=np.uint8)
    arr[rng.random(arr.shape) < 0.1] = 78
    save("dense_T8_S400", arr, loci_map(400, rng), synth.all_quartets(8), ref)

    # (2) tree-like simulated data, T=12, S=2000, all 495 quartets
    arr, tmap = synth.simulate_tmparr(12, 2000, seed=2)
    save("tree_T12_S2000", arr, tmap, synth.all_quartets(12), ref)

    # (3) sparse / degenerate: few informative sites, low substitution rate
    arr, tmap = synth.simulate_tmparr(10, 257, seed=3, p=0.01, missing=0.3)
    save("sparse_T10_S257", arr, tmap, synth.all_quartets(10), ref)

    # (4)+(5) zero-data quartets and an all-missing taxon
    rng = np.random.default_rng(4)
    arr = rng.integers(0, 4, size=(7, 130), dtype=np.uint8)
    arr[3, :] = 78                       # taxon 3 entirely missing
    arr[5, :] = arr[4, :]                # taxa 4,5 identical
    arr[6, :] = arr[4, :]                # ... and 6: quartets within {4,5,6,x} still vary via x
    arr[0, :] = 2                        # taxon 0 constant
    save("edge_T7_S130", arr, loci_map(130, rng), synth.all_quartets(7), ref)

    # (6) locus runs whose leading sites are all masked (subsample carry), long loci,
    #     plus a run crossing the 64- and 1024-site boundaries
    rng = np.random.default_rng(6)
    S = 2500
    arr = rng.integers(0, 4, size=(6, S), dtype=np.uint8)
    arr[rng.random(arr.shape) < 0.45] = 78
    tmap = loci_map(S, rng, mean=40)
    save("carry_T6_S2500", arr, tmap, synth.all_quartets(6), ref)

    # (7) tiny S (shorter than one wavefront) and S == 1
    rng = np.random.default_rng(7)
    arr = rng.integers(0, 4, size=(5, 37), dtype=np.uint8)
    save("tiny_T5_S37", arr, loci_map(37, rng, mean=2), synth.all_quartets(5), ref)
    arr = np.array([[0], [1], [0], [1], [2]], dtype=np.uint8)
    save("one_site_T5_S1", arr, loci_map(1, rng), synth.all_quartets(5), ref)

    # (8) low-rank: only two alleles segregating -> rank << 10 (minrank < 10 branch)
    rng = np.random.default_rng(8)
    arr = (rng.random((9, 700)) < 0.5).astype(np.uint8) * 2     # only A / G
    arr[rng.random(arr.shape) < 0.05] = 78
    save("lowrank_T9_S700", arr, loci_map(700, rng), synth.all_quartets(9), ref)

    # (11) the minrank < 10 branch (resolve_quartets.py:246) on rows whose argmin is NOT decided by rounding noise:
    #      sparse tree-like data, 14 taxa, all 1001 quartets -- 776 (full) / 853 (subsample) rows with a numerical rank
    #      below 10 and the two lowest scores well apart (sparse_T10_S257 has 377; lowrank_T9_S700 none: all of its rows are
    #      within 1e-9 * sigma_max in the reference itself)
    arr, tmap = synth.simulate_tmparr(14, 600, seed=9, p=0.01, missing=0.3)
    save("minrank_T14_S600", arr, tmap, synth.all_quartets(14), ref)

    # (9) config c1: 16 taxa, 5k SNPs, all 1820 quartets (BASELINE.json configs[0]).
    arr, tmap, qrts = synth.make_config("c1")
    save("c1_T16_S5000", arr, tmap, qrts, ref, with_cmats=False)

    make_slices(ref)
    make_rows(ref)


def load_reference_jitted():
    """tetrad/src/jitted.py (self-contained duplicate of tetrad/jit/*), numba.njit = identity.
    Its np.random.* calls therefore run on NumPy's legacy global RNG, not on numba's stream."""
    numba = types.ModuleType("numba")
    numba.njit = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules.setdefault("numba", numba)
    spec = importlib.util.spec_from_file_location("ref_jitted", "/root/reference/tetrad/src/jitted.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_resample_golden():
    """Bootstrap-replicate producer (run_inference.py:99-143 pieces).  The span table is
    deterministic; the two randomised functions are stored as run under NumPy's legacy RNG."""
    ref = load_reference_jitted()
    rng = np.random.default_rng(31)
    T, S = 7, 300
    alphabet = np.array([65, 67, 71, 84, 78, 82, 75, 83, 89, 87, 77], np.uint8)
    seqarr = alphabet[rng.choice(len(alphabet), size=(T, S), p=[.2, .2, .2, .2, .08, .02, .02, .02, .02, .02, .02])]
    tmap = loci_map(S, rng, mean=3)
    spans = ref.jit_get_spans(tmap.astype(np.int64))
    nloci = spans.shape[0]
    lidxs = rng.choice(nloci, nloci, replace=True)
    tmparr, tmpmap = ref.jit_resample(seqarr, spans, lidxs, seed=12345)
    resolved = ref.jit_resolve_ambigs(tmparr.copy(), seed=6789)
    np.savez_compressed(OUT / "resample_T7_S300.npz", seqarr=seqarr, maparr=tmap, spans=spans, lidxs=lidxs,
                        seed_resample=np.array(12345), seed_ambig=np.array(6789), tmparr=tmparr, tmpmap=tmpmap,
                        resolved=resolved, numpy_version=np.array(np.__version__))
    print("wrote resample_T7_S300.npz nloci", nloci, "S_out", tmparr.shape[1])
    # the toy map of the docstring at jit/get_spans.py:16-18 / deprecated/jitted.py:331-337
    toy = np.zeros((20, 2), np.int64)
    toy[:, 0] = [0] * 4 + [1] * 6 + [2] * 1 + [5] * 9
    toy[:, 1] = np.arange(20)
    np.savez_compressed(OUT / "spans_toy.npz", maparr=toy, spans=ref.jit_get_spans(toy))


def make_c5_golden():
    """BASELINE.json configs[4] shape: ONE bootstrap replicate of the c5 source made by the REFERENCE's
    resampler (jit_resample + jit_resolve_ambigs run under NumPy's legacy RNG, draws on the project
    Generator in the order of run_inference.py:117-123, recode as :133-136), then the reference worker on
    a few quartets of it.  The replicate itself (6.4 MB) is not stored: tests rebuild it with
    oracle/resample.py, which is pinned bit-for-bit against the same reference code, and check its CRC."""
    import zlib
    refj = load_reference_jitted()
    ref = load_reference()
    seqarr, maparr, spans = synth.make_c5_source()
    rng = np.random.default_rng(synth.CONFIG_SEEDS["c5"])
    nloci = spans.shape[0]
    assert np.array_equal(refj.jit_get_spans(maparr.astype(np.int64)), spans)
    lidxs = rng.choice(nloci, nloci, replace=True)                       # run_inference.py:117
    seed1 = int(rng.integers(2**31))
    tmparr, tmpmap = refj.jit_resample(seqarr, spans, lidxs, seed=seed1)  # :120
    seed2 = int(rng.integers(2**31))
    tmparr = refj.jit_resolve_ambigs(tmparr, seed=seed2)                  # :123
    tmparr[tmparr == 65] = 0                                              # :133-136
    tmparr[tmparr == 67] = 1
    tmparr[tmparr == 71] = 2
    tmparr[tmparr == 84] = 3
    qrts = synth.random_quartets(seqarr.shape[0], 96, seed=778)
    blob = dict(quartets=qrts, lidxs=lidxs, seed_resample=np.array(seed1), seed_ambig=np.array(seed2),
                replicate_shape=np.array(tmparr.shape), replicate_crc32=np.array(zlib.crc32(tmparr.tobytes())),
                tmpmap_crc32=np.array(zlib.crc32(np.ascontiguousarray(tmpmap).tobytes())),
                numpy_version=np.array(np.__version__))
    for sub in (False, True):
        res = run_reference(ref, tmparr, tmpmap, qrts, sub, with_cmats=True)
        tag = "sub" if sub else "full"
        for k, v in res.items():
            blob[f"{tag}_{k}"] = v
    np.savez_compressed(OUT / "c5_replicate_slice.npz", **blob)
    print("wrote c5_replicate_slice.npz", tmparr.shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["main", "resample", "c5"]
    if "main" in which:
        main()
    if "minrank" in which:
        main(only="minrank_T14_S600")
    if "slices" in which:
        main(only="slices")
    if "rows" in which:
        main(only="rows")
    if "resample" in which:
        make_resample_golden()
    if "c5" in which:
        make_c5_golden()
