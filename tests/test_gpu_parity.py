"""GPU parity tests proper: the HIP path, called through the C ABI, against the golden
vectors (reference outputs) and against the oracle on seeded inputs.

Bars: bit-exact for integer outputs (count matrices, nsnps, rank, topology on
non-degenerate quartets); singular values and scores within
|x - ref| <= 1e-6*|ref| + 1e-12*sigma_max  (north_star: 1e-6 relative; the absolute term
only covers values that are numerically zero, where a relative error is undefined).
"""
import numpy as np
import pytest

from conftest import GOLDEN_FULL_CASES, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-6
ATOL_REL_SMAX = 1e-12


@pytest.fixture(scope="module", params=["hqr", "jacobi", "hqr-one-quartet-kernels"])
def engine(request):
    """Every parity test runs against both device SVD paths: Householder + bidiagonal QR
    (default) and one-sided Jacobi -- and against both scan kernels: the "hqr" engine sends every batch of 64 quartets or
    more to the cooperative workgroup kernels (the default threshold is 4 096: smaller calls are latency-bound and go to the
    one-wave-per-quartet kernel) -- subsample mode to the plane-record scan (scan_f4.hpp), full mode to the joint-histogram scan
    (scan_dp.hpp) -- the "jacobi" engine keeps the defaults, so the small golden cases cover the one-wave kernel, and the third
    engine runs the cooperative kernel of rounds 1-3 in both modes."""
    from tetrad_amd.engine import QuartetEngine
    eng = QuartetEngine(0)
    eng.set_option("svd_method", 0 if request.param == "jacobi" else 1)
    if request.param == "hqr-one-quartet-kernels":
        # the cooperative kernel on nibble codes + plane records (tq_scan_wg_kernel: the default of rounds 1-3 and still that of
        # full-mode batches under 32 768 quartets) in both modes: no joint histogram, no plane-record-only scan
        eng.set_option("svd_method", 1)
        eng.set_option("wg_min_quartets", 64)
        eng.set_option("scan_dp", 0)
        eng.set_option("scan_f4", 0)
    if request.param == "hqr":
        eng.set_option("wg_min_quartets", 64)
        # ... and every full-mode batch to the joint-histogram scan (scan_dp.hpp; by default from 32 768 quartets on)
        eng.set_option("dp_min_quartets", 2)
    yield eng
    eng.close()


def assert_close(x, ref, smax, what):
    tol = RTOL * np.abs(ref) + ATOL_REL_SMAX * smax
    bad = np.abs(x - ref) > tol
    assert not bad.any(), f"{what}: {bad.sum()} values out of tolerance, worst {np.abs(x - ref).max()}"


def check_against(g, mode, rstat, rscor, flags, dbg):
    zero = g[f"{mode}_zero_data"]
    ref_rstat, ref_rscor = g[f"{mode}_rstat"], g[f"{mode}_rscor"]
    np.testing.assert_array_equal(rstat[:, 1], ref_rstat[:, 1])
    np.testing.assert_array_equal((flags & 1).astype(bool), zero)
    np.testing.assert_array_equal(rscor[zero], ref_rscor[zero])            # 0.001 rows
    assert (rstat[zero, 0] == 0).all()
    if f"{mode}_cmats" in g:
        np.testing.assert_array_equal(dbg["cmats"], g[f"{mode}_cmats"])
        ref_sv = g[f"{mode}_svds"]
        smax = ref_sv.max(axis=(1, 2))
        assert_close(dbg["svds"][~zero], ref_sv[~zero], smax[~zero, None, None], "singular values")
        ref_rank = (ref_sv > ref_sv.max(axis=2, keepdims=True) * 16 * np.finfo(float).eps).sum(axis=2)
        np.testing.assert_array_equal(dbg["ranks"][~zero], ref_rank[~zero])
    else:
        smax = np.full(len(rstat), np.abs(ref_rscor).max())
    assert_close(rscor[~zero], ref_rscor[~zero], smax[~zero, None], "scores")
    # topology: bit exact wherever the argmin is not decided by SVD rounding noise
    s = np.sort(ref_rscor, axis=1)
    ref_deg = (s[:, 1] - s[:, 0]) <= 1e-9 * smax
    dev_deg = (flags & 2) != 0
    if f"{mode}_cmats" in g:
        # the device may only excuse itself where the reference's own argmin is noise-decided: every row it
        # flags must have a reference gap within twice the flag threshold (an over-eager flag would hide a
        # wrong argmin)
        over = dev_deg & ~zero & ((s[:, 1] - s[:, 0]) > 2e-9 * smax)
        assert not over.any(), f"device flags {int(over.sum())} rows whose reference scores are well separated"
    assert ((flags & 8) == 0).all(), "singular-value iteration hit its sweep cap"
    ok = ~zero & ~ref_deg & ~dev_deg
    np.testing.assert_array_equal(rstat[ok, 0], ref_rstat[ok, 0])
    return int(ok.sum()), int((~zero & ~ok).sum())


@pytest.mark.parametrize("case", GOLDEN_FULL_CASES)
@pytest.mark.parametrize("mode", ["full", "sub"])
def test_golden_cases(engine, case, mode):
    g = load_golden(case)
    engine.set_data(g["tmparr"], g["tmpmap"])
    rstat, rscor, flags, dbg = engine.resolve(g["quartets"], mode == "sub", debug=True)
    check_against(g, mode, rstat, rscor, flags, dbg)


@pytest.mark.parametrize("mode", ["full", "sub"])
def test_c1_full_run(engine, mode):
    """BASELINE.json configs[0]: 16 taxa / 5k SNPs / all 1820 quartets vs the reference."""
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    rstat, rscor, flags = engine.resolve(g["quartets"], mode == "sub")
    nok, ndeg = check_against(g, mode, rstat, rscor, flags, None)
    assert ndeg == 0 and nok == 1820
    np.testing.assert_array_equal(rstat, g[f"{mode}_rstat"])


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_config_slices(engine, cfg):
    """Reference outputs for a few quartets of the c2 / c3 / c4 benchmark inputs (regenerated from the seed)."""
    from tetrad_amd import synth
    g = load_golden(f"{cfg}_slice")
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    engine.set_data(tmparr, tmpmap)
    for mode in ("full", "sub"):
        rstat, rscor, flags, dbg = engine.resolve(g["quartets"], mode == "sub", debug=True)
        nok, ndeg = check_against(g, mode, rstat, rscor, flags, dbg)
        assert ndeg == 0


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_reference_rows_at_benchmark_size(engine, cfg):
    """The REFERENCE worker's rows for 3 000 (c2) / 2 000 (c3) / 1 000 (c4) random quartets of the benchmark inputs
    (`c*_rows.npz`, made by the reference itself): site counts exact, scores within tolerance, topology exact on every row
    that neither the reference's scores nor the device flag as noise-decided -- and none is, on these inputs."""
    from tetrad_amd import synth
    g = load_golden(f"{cfg}_rows")
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    engine.set_data(tmparr, tmpmap)
    for mode in ("full", "sub"):
        rstat, rscor, flags = engine.resolve(g["quartets"], mode == "sub")
        nok, ndeg = check_against(g, mode, rstat, rscor, flags, None)
        assert ndeg == 0 and nok == len(g["quartets"])


@pytest.mark.parametrize("seed,T,S,missing,p", [(21, 20, 3000, 0.1, 0.05), (22, 9, 4097, 0.5, 0.02),
                                                 (23, 30, 2048, 0.0, 0.2), (24, 6, 65, 0.2, 0.1)])
def test_seeded_vs_oracle(engine, oracle, seed, T, S, missing, p):
    from tetrad_amd import synth
    tmparr, tmpmap = synth.simulate_tmparr(T, S, seed, p=p, missing=missing)
    rng = np.random.default_rng(seed)
    quartets = synth.all_quartets(T)
    quartets = quartets[rng.permutation(len(quartets))[:400]]
    engine.set_data(tmparr, tmpmap)
    for sub in (False, True):
        rstat, rscor, flags, dbg = engine.resolve(quartets, sub, debug=True)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, quartets, sub, debug=True)
        np.testing.assert_array_equal(dbg["cmats"], o["cmats"])
        np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
        zero = (o["flags"] & 1).astype(bool)
        np.testing.assert_array_equal((flags & 1).astype(bool), zero)
        smax = np.maximum(o["svds"].max(axis=(1, 2)), 1e-300)
        assert_close(dbg["svds"][~zero], o["svds"][~zero], smax[~zero, None, None], "singular values")
        assert_close(rscor, o_rscor, smax[:, None], "scores")
        np.testing.assert_array_equal(dbg["ranks"][~zero], o["rank"][~zero])
        ok = ((flags | o["flags"]) & 3) == 0
        np.testing.assert_array_equal(rstat[ok, 0], o_rstat[ok, 0])


@pytest.mark.parametrize("nrep", [1, 2, 4, 8, 16, 32])
def test_histogram_replica_counts_agree(engine, nrep):
    g = load_golden("tree_T12_S2000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    engine.set_option("nrep", nrep)
    try:
        for mode in ("full", "sub"):
            _, _, _, dbg = engine.resolve(g["quartets"][:97], mode == "sub", debug=True)
            np.testing.assert_array_equal(dbg["cmats"], g[f"{mode}_cmats"][:97])
    finally:
        engine.set_option("nrep", 0)


def test_locus_carry_across_lane_and_tile_boundaries(engine, oracle):
    """Subsample carry: loci longer than a lane's 32 sites and than a 2048-site tile, with
    leading sites of a run masked, and a run straddling every kind of boundary."""
    rng = np.random.default_rng(5)
    S = 3 * 2048 + 77
    arr = rng.integers(0, 4, size=(5, S), dtype=np.uint8)
    arr[rng.random(arr.shape) < 0.6] = 78
    lens = rng.choice([1, 2, 31, 32, 33, 64, 500, 2047, 2049], size=200)
    loc = np.repeat(np.arange(200, dtype=np.uint32), lens)[:S]
    loc = np.concatenate([loc, np.full(S - len(loc), 999, np.uint32)]) if len(loc) < S else loc
    tmap = np.stack([loc, np.arange(S, dtype=np.uint32)], axis=1)
    from tetrad_amd import synth
    quartets = synth.all_quartets(5)
    engine.set_data(arr, tmap)
    _, _, _, dbg = engine.resolve(quartets, True, debug=True)
    _, _, _, o = oracle.new_infer_resolved_quartets(arr, tmap, quartets, True, debug=True)
    np.testing.assert_array_equal(dbg["cmats"], o["cmats"])


def test_error_behaviour(oracle):
    from tetrad_amd.engine import QuartetEngine, TetradHipError
    with QuartetEngine(0) as eng:
        with pytest.raises(TetradHipError) as e:
            eng.resolve(np.array([[0, 1, 2, 3]], np.uint32), False)
        assert e.value.code == -4                                   # TQ_ERR_NO_DATA
        arr = np.zeros((5, 10), np.uint8)
        tmap = np.zeros((10, 2), np.uint32)
        tmap[:, 0] = [0, 0, 1, 1, 0, 0, 2, 2, 3, 3]                 # locus 0 re-appears
        eng.set_data(arr, tmap)
        with pytest.raises(TetradHipError) as e:
            eng.resolve(np.array([[0, 1, 2, 9]], np.uint32), False)
        assert e.value.code == -1                                   # taxon index >= T
        with pytest.raises(TetradHipError) as e:
            eng.resolve(np.array([[0, 1, 2, 3]], np.uint32), True)
        assert e.value.code == -5                                   # TQ_ERR_LOCUS_ORDER
        rstat, rscor, flags = eng.resolve(np.array([[0, 1, 2, 3]], np.uint32), False)
        assert flags[0] & 1 and rstat[0, 1] == 0 and np.all(rscor == 0.001)
        rstat, rscor, flags = eng.resolve(np.zeros((0, 4), np.uint32), False)
        assert rstat.shape == (0, 2)
        # a large batch is validated on the host while the GPU already works on it: same error, and the
        # context stays usable
        big = np.tile(np.array([[0, 1, 2, 3]], np.uint32), (70_000, 1))
        big[-1, 2] = 5
        with pytest.raises(TetradHipError) as e:
            eng.resolve(big, False)
        assert e.value.code == -1 and "69999" in str(e.value)
        big[-1, 2] = 2
        rstat, rscor, flags = eng.resolve(big, False)
        assert (flags & 1).all() and (rstat[:, 1] == 0).all()


def test_device_pointer_api_and_unranking(engine):
    """tq_resolve_dev / tq_resolve_range_dev / tq_unrank_dev agree with the host-buffer API."""
    import torch
    from tetrad_amd import synth
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    Q = 1820
    dev = torch.device("cuda:0")
    dq = torch.zeros((Q, 4), dtype=torch.int32, device=dev)
    rstat = torch.zeros((Q, 2), dtype=torch.int32, device=dev)
    rscor = torch.zeros((Q, 3), dtype=torch.float64, device=dev)
    flags = torch.zeros(Q, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    engine.resolve_range_dev(0, Q, True, dq.data_ptr(), rstat.data_ptr(), rscor.data_ptr(),
                             flags.data_ptr(), stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dq.cpu().numpy().astype(np.uint32), g["quartets"])
    np.testing.assert_array_equal(rstat.cpu().numpy().astype(np.uint32), g["sub_rstat"])
    h_rstat, h_rscor, h_flags = engine.resolve(g["quartets"], True)
    np.testing.assert_array_equal(rscor.cpu().numpy(), h_rscor)
    # random ranks -> quartets
    ranks = np.random.default_rng(0).choice(1820, size=500, replace=False).astype(np.int64)
    dr = torch.from_numpy(ranks).to(dev)
    dq2 = torch.zeros((500, 4), dtype=torch.int32, device=dev)
    engine.unrank_dev(dr.data_ptr(), 500, dq2.data_ptr(), stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dq2.cpu().numpy().astype(np.uint32), synth.unrank_quartets(ranks, 16))
    rstat2 = torch.zeros((500, 2), dtype=torch.int32, device=dev)
    rscor2 = torch.zeros((500, 3), dtype=torch.float64, device=dev)
    engine.resolve_dev(dq2.data_ptr(), 500, True, rstat2.data_ptr(), rscor2.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rstat2.cpu().numpy().astype(np.uint32), g["sub_rstat"][ranks])


@pytest.mark.parametrize("T", [16, 64, 128, 256])
def test_device_unranking_equals_reference_fixture(engine, T):
    """tq_unrank_dev (tq_unrank_kernel) against the reference's own `_index_to_combination` outputs
    (combinations.py:94-106; tests/golden/host_f2f3.npz made by make_golden_host.py)."""
    import torch
    g = load_golden("host_f2f3")
    ranks, want = g[f"unrank_T{T}_ranks"], g[f"unrank_T{T}_quartets"]
    engine.set_data(np.zeros((T, 64), np.uint8), np.zeros((64, 2), np.uint32))      # the kernel takes T from the data
    d_r = torch.from_numpy(ranks.astype(np.int64)).to("cuda:0")
    d_q = torch.zeros((len(ranks), 4), dtype=torch.int32, device="cuda:0")
    engine.unrank_dev(d_r.data_ptr(), len(ranks), d_q.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_q.cpu().numpy().view(np.uint32), want)


def test_full_size_properties_c3(engine):
    """At BASELINE.json's c3 size (128 taxa, 50k SNPs): size-independent properties.

    (1) chunk invariance / determinism; (2) nsnps bounds; (3) relabelling symmetry: swapping
    the last two taxa of a quartet swaps flattenings 1 and 2 (and leaves 0), swapping the
    middle two swaps flattenings 0 and 1; (4) sum of the count matrix equals nsnps."""
    from tetrad_amd import synth
    tmparr, tmpmap = synth.simulate_tmparr(128, 50_000, synth.CONFIG_SEEDS["c3"])
    engine.set_data(tmparr, tmpmap)
    q = synth.random_quartets(128, 6000, seed=99)
    nloci = len(np.unique(tmpmap[:, 0]))
    for sub in (False, True):
        rstat, rscor, flags = engine.resolve(q, sub)
        a = [engine.resolve(q[i:i + 1111], sub) for i in range(0, len(q), 1111)]
        np.testing.assert_array_equal(rstat, np.concatenate([x[0] for x in a]))
        np.testing.assert_array_equal(rscor, np.concatenate([x[1] for x in a]))
        assert (flags == 0).all(), "benchmark data must have no degenerate / zero-data quartets"
        assert (rstat[:, 1] <= (nloci if sub else 50_000)).all() and (rstat[:, 1] > 0).all()
        # relabelling symmetry
        q_cd = q[:, [0, 1, 3, 2]]
        r2, s2, _ = engine.resolve(q_cd, sub)
        np.testing.assert_array_equal(r2[:, 1], rstat[:, 1])
        smax = np.abs(rscor).max(axis=1, keepdims=True) * 1e3
        assert_close(s2[:, [0, 2, 1]], rscor, smax, "scores under c<->d swap")
        np.testing.assert_array_equal(r2[:, 0], np.array([0, 2, 1])[rstat[:, 0]])
        q_bc = q[:, [0, 2, 1, 3]]
        r3, s3, _ = engine.resolve(q_bc, sub)
        assert_close(s3[:, [1, 0, 2]], rscor, smax, "scores under b<->c swap")
        np.testing.assert_array_equal(r3[:, 0], np.array([1, 0, 2])[rstat[:, 0]])
        _, _, _, dbg = engine.resolve(q[:200], sub, debug=True)
        np.testing.assert_array_equal(dbg["cmats"].sum(axis=(2, 3)), np.repeat(rstat[:200, 1:2], 3, axis=1))
    full = engine.resolve(q, False)[0][:, 1]
    sub_ = engine.resolve(q, True)[0][:, 1]
    assert (sub_ <= full).all()


def test_c3_sample_vs_oracle(engine, oracle):
    """2,500 random quartets of the c3 benchmark input, both modes, against the oracle: exact
    counts / topology, scores within tolerance, no flagged rows."""
    from tetrad_amd import synth
    tmparr, tmpmap = synth.simulate_tmparr(128, 50_000, synth.CONFIG_SEEDS["c3"])
    q = synth.random_quartets(128, 2500, seed=4321)
    engine.set_data(tmparr, tmpmap)
    for sub in (True, False):
        rstat, rscor, flags = engine.resolve(q, sub)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
        assert (flags == 0).all() and (o["flags"] == 0).all()
        np.testing.assert_array_equal(rstat, o_rstat)
        smax = o["svds"].max(axis=(1, 2))[:, None]
        assert_close(rscor, o_rscor, smax, "scores")


@pytest.mark.parametrize("Q", [64, 65, 255, 257, 1001, 2047, 4100])
def test_scan_grid_shapes(engine, Q):
    """Batch sizes that do not fill the last workgroup / the last XCD chunk: the cooperative kernel
    (one block per workgroup, XCD-contiguous ids) must give the rows of the one-wave-per-quartet
    kernel.  A full-mode pass runs in between so that a skipped block would show stale counts."""
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    rng = np.random.default_rng(Q)
    q = g["quartets"][rng.integers(0, len(g["quartets"]), size=Q)]
    try:
        engine.set_option("scan_wg", 1)
        want = engine.resolve(q, True)
        engine.set_option("scan_wg", 0)
        engine.resolve(q, False)
        got = engine.resolve(q, True)
    finally:
        engine.set_option("scan_wg", 0)
    for a, b in zip(want, got):
        np.testing.assert_array_equal(a, b)


def test_pipeline_options_do_not_change_results(engine):
    """Per-wave scan kernel, natural order, small scan batches, small singular-value chunks (= result-copy
    pieces of the host API): all must give bitwise the same rows as the default configuration."""
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    q = np.tile(g["quartets"], (3, 1))
    base = engine.resolve(q, True)
    defaults = {"scan_wg": 0, "xcd_remap": 1, "order": 1, "scan_method": -1, "waves_per_cu": 0, "svd_wpc": 0,
                "batch": 0, "svd_chunk": 0, "share_c": 0, "svd_streams": 0, "park_t": 1, "scan_pair": 0}
    try:
        for opts in ({"scan_wg": 1}, {"scan_wg": 2}, {"scan_wg": 8}, {"scan_wg": 16}, {"xcd_remap": 0},
                     {"waves_per_cu": 12}, {"svd_wpc": 8}, {"order": 0}, {"batch": 700}, {"scan_method": 0},
                     {"svd_chunk": 1000}, {"svd_chunk": 64, "batch": 999}, {"svd_chunk": 1}, {"share_c": 1},
                     {"share_c": 1, "scan_method": 0}, {"svd_streams": 1, "svd_chunk": 500}, {"park_t": 0},
                     {"park_t": 0, "scan_method": 1}, {"scan_method": 1}, {"scan_pair": 1}, {"scan_pair": 1, "scan_method": 0},
                     {"scan_pair": 1, "order": 0, "batch": 701}):
            for k, v in opts.items():
                engine.set_option(k, v)
            if opts.get("svd_chunk") == 1:
                got = engine.resolve(q[:40], True)
                want = tuple(x[:40] for x in base)
            else:
                got, want = engine.resolve(q, True), base
            for a, b in zip(want, got):
                np.testing.assert_array_equal(a, b)
            for k in opts:
                engine.set_option(k, defaults[k])
    finally:
        for k, v in defaults.items():
            engine.set_option(k, v)


def test_pageable_and_pinned_result_arrays_agree(engine):
    """tq_resolve writes page-locked result arrays with the copy engine directly and pageable ones through
    pinned staging pieces + host memcpy; tq_resolve_to_host takes the quartets from the device.  Several
    pieces per call (svd_chunk), last piece ragged."""
    import ctypes
    import torch
    from tetrad_amd.engine import pinned_empty
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    q = np.tile(g["quartets"], (4, 1))[:7001]
    Q = len(q)
    want = engine.resolve(q, True)
    lib, h = engine._lib, engine._h
    try:
        engine.set_option("svd_chunk", 1500)
        for pinned in (False, True):
            alloc = pinned_empty if pinned else (lambda shape, dt: np.full(shape, 7, dtype=dt))
            rstat, rscor, flags = alloc((Q, 2), np.uint32), alloc((Q, 3), np.float64), alloc(Q, np.uint8)
            rc = lib.tq_resolve(h, ctypes.c_void_p(q.ctypes.data), Q, 1, ctypes.c_void_p(rstat.ctypes.data),
                                ctypes.c_void_p(rscor.ctypes.data), ctypes.c_void_p(flags.ctypes.data))
            assert rc == 0
            for a, b in zip(want, (rstat, rscor, flags)):
                np.testing.assert_array_equal(a, b)
            # flags may be NULL
            rstat2, rscor2 = alloc((Q, 2), np.uint32), alloc((Q, 3), np.float64)
            rc = lib.tq_resolve(h, ctypes.c_void_p(q.ctypes.data), Q, 1, ctypes.c_void_p(rstat2.ctypes.data),
                                ctypes.c_void_p(rscor2.ctypes.data), None)
            assert rc == 0
            np.testing.assert_array_equal(rstat2, want[0])
            np.testing.assert_array_equal(rscor2, want[1])
        dq = torch.from_numpy(q.astype(np.int32)).to("cuda:0")
        got = engine.resolve_to_host(dq.data_ptr(), Q, True)
        for a, b in zip(want, got):
            np.testing.assert_array_equal(a, b)
        out = (np.zeros((Q, 2), np.uint32), np.zeros((Q, 3), np.float64), np.zeros(Q, np.uint8))
        engine.resolve_to_host(dq.data_ptr(), Q, True, out=out)
        for a, b in zip(want, out):
            np.testing.assert_array_equal(a, b)
    finally:
        engine.set_option("svd_chunk", 0)


@pytest.mark.parametrize("Q", [65536, 100_000, 262_144, 262_145, 300_001])
def test_pageable_results_mid_size_calls_default_chunking(engine, Q):
    """Pageable result arrays at the DEFAULT singular-value chunking, for call sizes around the point where
    stage_svd cuts one piece into two half-chunks on two streams (65 536 <= Q < 2 * svd_chunk): the drop-in
    binding of INTEGRATION.md hands np.zeros arrays to tq_resolve at exactly such sizes (the reference's
    get_chunksize gives 78 125-quartet chunks for 5e6 quartets on 4 cores, run_inference.py:73-96).
    Every row must arrive, in its own slot, == the page-locked path."""
    import ctypes
    import torch
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    base = g["quartets"]
    q = np.ascontiguousarray(np.tile(base, (Q // len(base) + 1, 1))[:Q])
    # make rows distinguishable by position: rotate the tiling so that row i != row i + Q/2 in general
    q = np.ascontiguousarray(np.roll(q, 37, axis=0))
    want = engine.resolve(q, True)                          # page-locked arrays (direct path)
    assert want[0][:, 1].max() > 0
    lib, h = engine._lib, engine._h
    rstat, rscor, flags = np.full((Q, 2), 7, np.uint32), np.full((Q, 3), 7.0), np.full(Q, 7, np.uint8)
    rc = lib.tq_resolve(h, ctypes.c_void_p(q.ctypes.data), Q, 1, ctypes.c_void_p(rstat.ctypes.data),
                        ctypes.c_void_p(rscor.ctypes.data), ctypes.c_void_p(flags.ctypes.data))
    assert rc == 0
    for a, b in zip(want, (rstat, rscor, flags)):
        np.testing.assert_array_equal(a, b)
    dq = torch.from_numpy(q.astype(np.int32)).to("cuda:0")
    out = (np.full((Q, 2), 9, np.uint32), np.full((Q, 3), 9.0), np.full(Q, 9, np.uint8))
    engine.resolve_to_host(dq.data_ptr(), Q, True, out=out)
    for a, b in zip(want, out):
        np.testing.assert_array_equal(a, b)


def test_scan_then_svd_in_pieces(engine):
    """tq_scan_dev + tq_svd_dev over arbitrary row ranges, pieces written to separate slabs, == tq_resolve."""
    import torch
    g = load_golden("c1_T16_S5000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    q = np.tile(g["quartets"], (2, 1))[:3001]
    Q = len(q)
    want = engine.resolve(q, False)
    dev = torch.device("cuda:0")
    dq = torch.from_numpy(q.astype(np.int32)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    engine.scan_dev(dq.data_ptr(), Q, False, stream)
    pieces = [(0, 1000), (1000, 1), (1001, 1999), (3000, 1)]
    outs = []
    for q0, n in reversed(pieces):                       # any order
        rs = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        sc = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        fl = torch.zeros(n, dtype=torch.uint8, device=dev)
        engine.svd_dev(q0, n, rs.data_ptr(), sc.data_ptr(), fl.data_ptr(), stream)
        outs.append((q0, rs, sc, fl))
    torch.cuda.synchronize()
    for q0, rs, sc, fl in outs:
        n = rs.shape[0]
        np.testing.assert_array_equal(rs.cpu().numpy().astype(np.uint32), want[0][q0:q0 + n])
        np.testing.assert_array_equal(sc.cpu().numpy(), want[1][q0:q0 + n])
        np.testing.assert_array_equal(fl.cpu().numpy(), want[2][q0:q0 + n])
    from tetrad_amd.engine import TetradHipError
    with pytest.raises(TetradHipError):
        engine.svd_dev(3000, 2, outs[0][1].data_ptr(), outs[0][2].data_ptr(), 0, stream)   # beyond the scanned batch


def test_sweep_cap_is_flagged(engine):
    """A singular-value iteration that hits its sweep cap must say so (numpy raises LinAlgError there):
    with the cap forced to one sweep per value most matrices do not converge -> TQ_FLAG_NO_CONVERGENCE and
    the Python mirror raises; with the default cap no row of any golden case carries the flag."""
    from tetrad_amd import resolve_quartets as RQ
    g = load_golden("tree_T12_S2000")
    engine.set_data(g["tmparr"], g["tmpmap"])
    _, _, flags = engine.resolve(g["quartets"], False)
    assert (flags & 8).sum() == 0
    try:
        engine.set_option("svd_method", 1)
        engine.set_option("bdsqr_maxit", 1)
        _, _, flags = engine.resolve(g["quartets"], False)
        assert (flags & 8).sum() > len(flags) // 2
    finally:
        engine.set_option("bdsqr_maxit", 0)
    eng0 = RQ.get_engine(0)
    try:
        eng0.set_option("bdsqr_maxit", 1)
        with pytest.raises(np.linalg.LinAlgError):
            RQ.new_infer_resolved_quartets(g["tmparr"], g["tmpmap"], g["quartets"], False)
    finally:
        eng0.set_option("bdsqr_maxit", 0)


@pytest.mark.parametrize("T,S,seed,p,missing,sub", [
    (30, 400, 77, 0.02, 0.35, True),      # sparse: many rank-deficient matrices, zero columns between live ones
    (24, 250, 5, 0.01, 0.60, False),      # very sparse, full mode
    (22, 3000, 9, 0.30, 0.00, True),      # saturated: every pattern occurs, full rank
])
def test_hqr_and_jacobi_agree_on_stress_data(oracle, T, S, seed, p, missing, sub):
    """The two device SVD paths against each other on every quartet of awkward inputs, and both
    against numpy's rank on a sample: identical ranks and nsnps, singular values within tolerance,
    identical topology wherever neither path flags the row.  (This test found the zero-column
    cascade in the Householder path that the golden cases did not exercise.)"""
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    tmparr, tmpmap = synth.simulate_tmparr(T, S, seed=seed, p=p, missing=missing)
    q = synth.all_quartets(T)
    res = {}
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        for name, m in (("hqr", 1), ("jacobi", 0)):
            eng.set_option("svd_method", m)
            res[name] = eng.resolve(q, sub, debug=True)
    (r1, s1, f1, d1), (r0, s0, f0, d0) = res["hqr"], res["jacobi"]
    np.testing.assert_array_equal(r1[:, 1], r0[:, 1])
    np.testing.assert_array_equal(d1["cmats"], d0["cmats"])
    np.testing.assert_array_equal(d1["ranks"], d0["ranks"])
    smax = np.maximum(d0["svds"].max(axis=(1, 2)), 1e-300)
    assert_close(d1["svds"], d0["svds"], smax[:, None, None], "singular values HQR vs Jacobi")
    assert_close(s1, s0, smax[:, None], "scores HQR vs Jacobi")
    assert np.isfinite(s1).all() and np.isfinite(d1["svds"]).all()
    ok = ((f1 | f0) & 3) == 0
    np.testing.assert_array_equal(r1[ok, 0], r0[ok, 0])
    # numpy.linalg.matrix_rank on a sample of the count matrices
    idx = np.random.default_rng(1).choice(len(q), 400, replace=False)
    for qi in idx:
        if r0[qi, 1] == 0:
            continue
        for t in range(3):
            assert np.linalg.matrix_rank(d0["cmats"][qi, t].astype(np.float64)) == d1["ranks"][qi, t]


@pytest.mark.parametrize("T,S,seed,p,missing", [(11, 700, 21, 0.01, 0.35), (12, 3000, 22, 0.05, 0.1), (10, 300, 23, 0.3, 0.0)])
def test_bidiag_layouts_agree(T, S, seed, p, missing):
    """The two lane layouts of the Householder bidiagonalisation (2 x 2 over the quad, the default, and four column groups)
    apply the same reflectors in a different summation order: identical ranks, nsnps and topologies on unflagged rows,
    singular values and scores to rounding, on sparse (rank-deficient), dense and saturated inputs; small batches (one block per
    flattening) and large ones."""
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    tmparr, tmpmap = synth.simulate_tmparr(T, S, seed=seed, p=p, missing=missing)
    q = synth.all_quartets(T)
    q = np.concatenate([q] * (1 + 40000 // len(q)))[:40000] if S == 3000 else q      # one case above the tsplit threshold
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        for sub in (True, False):
            res = {}
            for layout in (1, 0):
                eng.set_option("bidiag_layout", layout)
                res[layout] = eng.resolve(q[:3000], sub, debug=True) if len(q) > 3000 else eng.resolve(q, sub, debug=True)
                if len(q) > 3000:
                    res[layout, "big"] = eng.resolve(q, sub)
            eng.set_option("bidiag_layout", -1)
            (r1, s1, f1, d1), (r0, s0, f0, d0) = res[1], res[0]
            np.testing.assert_array_equal(r1[:, 1], r0[:, 1])
            np.testing.assert_array_equal(d1["ranks"], d0["ranks"])
            smax = np.maximum(d0["svds"].max(axis=(1, 2)), 1e-300)
            assert np.abs(d1["svds"] - d0["svds"]).max() <= 1e-11 * smax.max()
            assert_close(s1, s0, smax[:, None], "scores, 2 x 2 layout vs column layout")
            ok = ((f1 | f0) & 3) == 0
            np.testing.assert_array_equal(r1[ok, 0], r0[ok, 0])
            if len(q) > 3000:
                (rb1, sb1, fb1), (rb0, sb0, fb0) = res[1, "big"], res[0, "big"]
                np.testing.assert_array_equal(rb1[:, 1], rb0[:, 1])
                okb = ((fb1 | fb0) & 3) == 0
                np.testing.assert_array_equal(rb1[okb, 0], rb0[okb, 0])
                # chunk invariance across the tsplit threshold: the first 3000 rows of the big batch == the small batch
                np.testing.assert_array_equal(rb1[:3000], r1)
                np.testing.assert_array_equal(sb1[:3000], s1)


def test_unsorted_and_repeated_taxon_indices(engine, oracle):
    """The reference accepts any four row indices (order matters, repeats allowed); so must the engine."""
    from tetrad_amd import synth
    tmparr, tmpmap = synth.simulate_tmparr(11, 2500, seed=31, p=0.08, missing=0.15)
    rng = np.random.default_rng(8)
    q = rng.integers(0, 11, size=(600, 4)).astype(np.uint32)
    engine.set_data(tmparr, tmpmap)
    for sub in (True, False):
        rstat, rscor, flags, dbg = engine.resolve(q, sub, debug=True)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
        np.testing.assert_array_equal(dbg["cmats"], o["cmats"])
        np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
        np.testing.assert_array_equal((flags & 1).astype(bool), (o["flags"] & 1).astype(bool))
        smax = np.maximum(o["svds"].max(axis=(1, 2)), 1e-300)
        assert_close(rscor, o_rscor, smax[:, None], "scores")
        ok = ((flags | o["flags"]) & 3) == 0
        np.testing.assert_array_equal(rstat[ok, 0], o_rstat[ok, 0])


def test_many_taxa_small_sites(engine, oracle):
    """T = 300 taxa (row offsets beyond the c4 shape), S barely above one 2048-site step."""
    from tetrad_amd import synth
    tmparr, tmpmap = synth.simulate_tmparr(300, 2100, seed=300, p=0.03, missing=0.1)
    q = synth.random_quartets(300, 1500, seed=12)
    engine.set_data(tmparr, tmpmap)
    for sub in (True, False):
        rstat, rscor, flags = engine.resolve(q, sub)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
        np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
        smax = np.maximum(o["svds"].max(axis=(1, 2)), 1e-300)
        assert_close(rscor, o_rscor, smax[:, None], "scores")
        ok = ((flags | o["flags"]) & 3) == 0
        np.testing.assert_array_equal(rstat[ok, 0], o_rstat[ok, 0])


def test_taxa_beyond_three_field_sort_key(engine, oracle):
    """T = 1700: T^3 no longer fits the 32-bit sort key, so the order falls back to (a,b) only."""
    from tetrad_amd import synth
    tmparr, tmpmap = synth.simulate_tmparr(1700, 300, seed=17, p=0.02, missing=0.1)
    q = synth.random_quartets(1700, 1500, seed=3)
    engine.set_data(tmparr, tmpmap)
    for sub in (True, False):
        rstat, rscor, flags = engine.resolve(q, sub)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
        np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
        smax = np.maximum(o["svds"].max(axis=(1, 2)), 1e-300)
        assert_close(rscor, o_rscor, smax[:, None], "scores")
        ok = ((flags | o["flags"]) & 3) == 0
        np.testing.assert_array_equal(rstat[ok, 0], o_rstat[ok, 0])
