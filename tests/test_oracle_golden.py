"""Pin the oracle (oracle/) against outputs of the reference's own functions
(tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import GOLDEN_FULL_CASES, load_golden


@pytest.mark.parametrize("case", GOLDEN_FULL_CASES)
@pytest.mark.parametrize("mode", ["full", "sub"])
def test_oracle_matches_reference(oracle, case, mode):
    g = load_golden(case)
    sub = mode == "sub"
    q, rstat, rscor, dbg = oracle.new_infer_resolved_quartets(
        g["tmparr"], g["tmpmap"], g["quartets"], sub, debug=True)
    zero = g[f"{mode}_zero_data"]
    # integer outputs: bit exact
    np.testing.assert_array_equal(q, g["quartets"])
    np.testing.assert_array_equal(dbg["cmats"], g[f"{mode}_cmats"])
    np.testing.assert_array_equal(rstat[:, 1], g[f"{mode}_rstat"][:, 1])
    np.testing.assert_array_equal((dbg["flags"] & 1).astype(bool), zero)
    # same numpy.linalg calls as the reference => bit-identical here
    np.testing.assert_array_equal(rscor, g[f"{mode}_rscor"])
    np.testing.assert_array_equal(dbg["svds"][~zero], g[f"{mode}_svds"][~zero])
    # topology: everywhere except the reference's unseeded random pick
    np.testing.assert_array_equal(rstat[~zero, 0], g[f"{mode}_rstat"][~zero, 0])


@pytest.mark.parametrize("mode", ["full", "sub"])
def test_oracle_c1_full_run(oracle, mode):
    """BASELINE.json configs[0]: 16 taxa, 5k SNPs, all 1820 quartets."""
    g = load_golden("c1_T16_S5000")
    q, rstat, rscor = oracle.new_infer_resolved_quartets(
        g["tmparr"], g["tmpmap"], g["quartets"], mode == "sub")
    assert not g[f"{mode}_zero_data"].any()
    np.testing.assert_array_equal(rstat, g[f"{mode}_rstat"])
    np.testing.assert_array_equal(rscor, g[f"{mode}_rscor"])


def test_numpy_spelled_out_variant_matches(oracle):
    g = load_golden("dense_T8_S400")
    for sub in (False, True):
        a = oracle.new_infer_resolved_quartets(g["tmparr"], g["tmpmap"], g["quartets"], sub)
        b = oracle.new_infer_resolved_quartets_numpy(g["tmparr"], g["tmpmap"], g["quartets"], sub)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("mode", ["full", "sub"])
def test_tuned_cpu_variant_matches_reference_vectors(oracle, mode):
    """The batched values-only-SVD variant timed by bench.py as `cpu_baseline.tuned_variant`: same
    integers as the reference's c1 run, scores to rounding (it makes one LAPACK call where the
    reference makes two different ones)."""
    g = load_golden("c1_T16_S5000")
    q, rstat, rscor = oracle.new_infer_resolved_quartets_batched(
        g["tmparr"], g["tmpmap"], g["quartets"], mode == "sub", chunk=500)
    np.testing.assert_array_equal(rstat, g[f"{mode}_rstat"])
    np.testing.assert_allclose(rscor, g[f"{mode}_rscor"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("case", ["tiny_T5_S37", "edge_T7_S130", "carry_T6_S2500"])
def test_c_count_kernels_match_python_loops(oracle, case):
    g = load_golden(case)
    arr, loc = g["tmparr"], np.ascontiguousarray(g["tmpmap"][:, 0])
    for qd in g["quartets"][:6]:
        seqs = arr[qd, :]
        mask = np.sum(seqs >= 78, axis=0) + (np.sum(seqs == seqs[0], axis=0) == 4)
        for sub in (False, True):
            fn = oracle.subsample_chunk_to_matrices if sub else oracle.full_chunk_to_matrices
            np.testing.assert_array_equal(
                fn(seqs, loc, mask), oracle.chunk_to_matrices_py(seqs, loc, mask, sub))


def test_flattenings_are_tensor_transposes(oracle):
    """M1 = C.transpose(0,2,1,3), M2 = C.transpose(0,3,1,2)."""
    g = load_golden("tree_T12_S2000")
    cm = g["full_cmats"][:50]
    C = cm[:, 0].reshape(-1, 4, 4, 4, 4)
    np.testing.assert_array_equal(cm[:, 1], C.transpose(0, 1, 3, 2, 4).reshape(-1, 16, 16))
    np.testing.assert_array_equal(cm[:, 2], C.transpose(0, 1, 4, 2, 3).reshape(-1, 16, 16))


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_oracle_matches_reference_on_config_slices(oracle, cfg):
    """Reference outputs for a few quartets of the BASELINE.json benchmark inputs (regenerated from the seed)."""
    from tetrad_amd import synth
    g = load_golden(f"{cfg}_slice")
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    for mode in ("full", "sub"):
        _, rstat, rscor, dbg = oracle.new_infer_resolved_quartets(tmparr, tmpmap, g["quartets"], mode == "sub", debug=True)
        assert not g[f"{mode}_zero_data"].any()
        np.testing.assert_array_equal(dbg["cmats"], g[f"{mode}_cmats"])
        np.testing.assert_array_equal(rstat, g[f"{mode}_rstat"])
        np.testing.assert_array_equal(rscor, g[f"{mode}_rscor"])


@pytest.mark.parametrize("cfg,n", [("c2", 300), ("c3", 150), ("c4", 60)])
def test_oracle_matches_reference_rows_at_benchmark_size(oracle, cfg, n):
    """`c{2,3,4}_rows.npz`: the reference worker's rows for 3 000 / 2 000 / 1 000 quartets of the benchmark inputs (the GPU
    test compares all of them with the device); the oracle is checked on the first n of each here (CPU time)."""
    from tetrad_amd import synth
    g = load_golden(f"{cfg}_rows")
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    for mode in ("full", "sub"):
        _, rstat, rscor = oracle.new_infer_resolved_quartets(tmparr, tmpmap, g["quartets"][:n], mode == "sub")
        np.testing.assert_array_equal(rstat, g[f"{mode}_rstat"][:n])
        np.testing.assert_array_equal(rscor, g[f"{mode}_rscor"][:n])


def test_oracle_matches_reference_on_c5_replicate(oracle):
    """BASELINE.json configs[4]: the bootstrap replicate the reference's resampler made from the c5 source is
    rebuilt by the oracle's restatement (CRC-checked), and the oracle worker reproduces the reference's rows."""
    import zlib
    from oracle import resample as R
    from tetrad_amd import synth
    g = load_golden("c5_replicate_slice")
    seqarr, maparr, spans = synth.make_c5_source()
    np.testing.assert_array_equal(R.get_spans(maparr), spans)
    tmparr, tmpmap = R.resample(seqarr, spans, g["lidxs"], seed=int(g["seed_resample"]))
    tmparr = R.recode(R.resolve_ambigs(tmparr, seed=int(g["seed_ambig"])))
    assert zlib.crc32(tmparr.tobytes()) == int(g["replicate_crc32"])
    assert zlib.crc32(np.ascontiguousarray(tmpmap).tobytes()) == int(g["tmpmap_crc32"])
    # the same draws on the project Generator (run_inference.py:117-123)
    rng = np.random.default_rng(synth.CONFIG_SEEDS["c5"])
    np.testing.assert_array_equal(rng.choice(len(spans), len(spans), replace=True), g["lidxs"])
    assert int(rng.integers(2**31)) == int(g["seed_resample"]) and int(rng.integers(2**31)) == int(g["seed_ambig"])
    for mode in ("full", "sub"):
        _, rstat, rscor, dbg = oracle.new_infer_resolved_quartets(tmparr, tmpmap, g["quartets"], mode == "sub", debug=True)
        np.testing.assert_array_equal(dbg["cmats"], g[f"{mode}_cmats"])
        np.testing.assert_array_equal(rstat, g[f"{mode}_rstat"])
        np.testing.assert_array_equal(rscor, g[f"{mode}_rscor"])


def test_minrank_fixture_covers_the_low_rank_branch_with_decided_rows():
    """What `minrank_T14_S600` is there for: resolve_quartets.py:246 takes `minrank = min(10, rank.min())`; the fixture
    must hold many rows whose numerical rank is below 10 AND whose two lowest reference scores are well apart (so that
    the topology comparison on them means something) -- counted from the reference's own singular values."""
    g = load_golden("minrank_T14_S600")
    for mode, at_least in (("full", 700), ("sub", 800)):
        sv, rscor, zero = g[f"{mode}_svds"], g[f"{mode}_rscor"], g[f"{mode}_zero_data"]
        rank = (sv > sv.max(axis=2, keepdims=True) * 16 * np.finfo(float).eps).sum(axis=2).min(axis=1)
        s = np.sort(rscor, axis=1)
        decided = (s[:, 1] - s[:, 0]) > 2e-9 * sv.max(axis=(1, 2))
        n = int(((rank < 10) & decided & ~zero).sum())
        assert n >= at_least, (mode, n)
