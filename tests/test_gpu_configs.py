"""GPU: BASELINE.json configs[3] (c4: 256 taxa x 100k SNPs, 5e6-quartet batch) and configs[4] (c5: 128 taxa,
~50k SNPs per bootstrap replicate) at full size, and the drop-in surface of INTEGRATION.md section 1
(`tetrad_amd.resolve_quartets`), called the way the reference's distributor calls it
(run_inference.py:219-220)."""
import zlib
from itertools import combinations, islice

import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import assert_close, check_against

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from tetrad_amd.engine import QuartetEngine
    eng = QuartetEngine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def c4_data():
    from tetrad_amd import synth
    T, S, _ = synth.CONFIGS["c4"]
    return synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS["c4"])


def test_c4_full_batch_properties_and_oracle_sample(engine, oracle, c4_data):
    """The whole c4 batch -- 5e6 random quartets of 256 x 100k in ONE scan batch (5 GB count slab), results
    delivered to host arrays in pieces -- then size-independent properties and an oracle sample:
    (1) rows of the big batch == the same quartets resolved alone (batch invariance, bitwise);
    (2) no flagged row, 0 < nsnps <= nloci; (3) relabelling symmetry; (4) oracle on 300 rows, both modes."""
    import torch
    from math import comb
    from tetrad_amd import synth
    tmparr, tmpmap = c4_data
    T, S = tmparr.shape
    Q = 5_000_000
    engine.set_data(tmparr, tmpmap)
    ranks = np.random.default_rng(synth.CONFIG_SEEDS["c4"] + 1000).choice(comb(T, 4), size=Q, replace=False)
    dev = torch.device("cuda:0")
    d_r = torch.from_numpy(ranks.astype(np.int64)).to(dev)
    d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    engine.unrank_dev(d_r.data_ptr(), Q, d_q.data_ptr(), stream)
    torch.cuda.synchronize()
    rstat, rscor, flags = engine.resolve_to_host(d_q.data_ptr(), Q, True)
    nloci = len(np.unique(tmpmap[:, 0]))
    assert (flags == 0).all(), "benchmark data must have no degenerate / zero-data / unconverged rows"
    assert (rstat[:, 1] > 0).all() and (rstat[:, 1] <= nloci).all() and (rstat[:, 0] <= 2).all()
    pick = np.random.default_rng(5).choice(Q, size=3000, replace=False)
    pick[:3] = [0, Q - 1, Q // 2]
    q = synth.unrank_quartets(ranks[pick], T)
    np.testing.assert_array_equal(d_q[torch.from_numpy(pick).to(dev)].cpu().numpy().view(np.uint32), q)
    r1, s1, f1 = engine.resolve(q, True)
    np.testing.assert_array_equal(r1, rstat[pick])
    np.testing.assert_array_equal(s1, rscor[pick])
    # relabelling symmetry at this size (swapping taxa c<->d swaps flattenings 1 and 2)
    r2, s2, _ = engine.resolve(q[:, [0, 1, 3, 2]], True)
    np.testing.assert_array_equal(r2[:, 1], r1[:, 1])
    np.testing.assert_array_equal(r2[:, 0], np.array([0, 2, 1])[r1[:, 0]])
    assert_close(s2[:, [0, 2, 1]], s1, np.abs(s1).max(axis=1, keepdims=True) * 1e3, "scores under c<->d swap")
    for sub in (True, False):
        got = engine.resolve(q[:300], sub)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q[:300], sub, debug=True)
        assert (got[2] == 0).all() and (o["flags"] == 0).all()
        np.testing.assert_array_equal(got[0], o_rstat)
        assert_close(got[1], o_rscor, o["svds"].max(axis=(1, 2))[:, None], "scores")


def test_c4_batch_and_chunk_invariance(engine, c4_data):
    """>= 1e5 quartets of c4: scan batches of 40k, singular-value chunks of 7k and the single-stream stage give
    bitwise the rows of the default pass; full mode too."""
    from tetrad_amd import synth
    tmparr, tmpmap = c4_data
    engine.set_data(tmparr, tmpmap)
    q = synth.random_quartets(256, 120_000, seed=44)
    for sub in (True, False):
        base = engine.resolve(q, sub)
        assert (base[2] == 0).all()
        try:
            for opts in ({"batch": 40_000}, {"svd_chunk": 7_000}, {"svd_streams": 1, "svd_chunk": 50_000}):
                for k, v in opts.items():
                    engine.set_option(k, v)
                got = engine.resolve(q, sub)
                for a, b in zip(base, got):
                    np.testing.assert_array_equal(a, b)
                for k in opts:
                    engine.set_option(k, 0)
        finally:
            for k in ("batch", "svd_chunk", "svd_streams"):
                engine.set_option(k, 0)


def _c5_reference_replicate():
    """The replicate the golden file was made from, rebuilt with the oracle's restatement of the reference's
    resampler (pinned bit-for-bit against the reference's code) and checked against the stored CRCs."""
    from oracle import resample as R
    from tetrad_amd import synth
    g = load_golden("c5_replicate_slice")
    seqarr, maparr, spans = synth.make_c5_source()
    tmparr, tmpmap = R.resample(seqarr, spans, g["lidxs"], seed=int(g["seed_resample"]))
    tmparr = R.recode(R.resolve_ambigs(tmparr, seed=int(g["seed_ambig"])))
    assert tuple(g["replicate_shape"]) == tmparr.shape
    assert zlib.crc32(tmparr.tobytes()) == int(g["replicate_crc32"])
    assert zlib.crc32(np.ascontiguousarray(tmpmap).tobytes()) == int(g["tmpmap_crc32"])
    return g, seqarr, spans, tmparr, tmpmap


def test_c5_reference_replicate_slice(engine):
    """c5 shape: a bootstrap replicate made by the reference's own resampler, reference outputs for 96 quartets."""
    g, _, _, tmparr, tmpmap = _c5_reference_replicate()
    engine.set_data(tmparr, tmpmap)
    for mode in ("full", "sub"):
        rstat, rscor, flags, dbg = engine.resolve(g["quartets"], mode == "sub", debug=True)
        nok, ndeg = check_against(g, mode, rstat, rscor, flags, dbg)
        assert ndeg == 0


def test_c5_size_device_replicate_vs_oracle(engine, oracle):
    """One c5-size replicate built ON THE DEVICE (128 taxa, ~50k sites: locus resample + shuffle + IUPAC
    resolution), exported, structurally valid for a sample of loci, and 2,500 quartets resolved on it
    == the oracle on the exported replicate."""
    from oracle import resample as R
    from tetrad_amd import bootstrap, synth
    seqarr, maparr, spans = synth.make_c5_source()
    engine.set_source(seqarr, spans)
    rng = np.random.default_rng(synth.CONFIG_SEEDS["c5"])
    lidxs, s1, s2 = bootstrap.draw_replicate(len(spans), rng)
    S = engine.bootstrap(lidxs, s1, s2)
    tmparr, tmpmap = engine.get_data()
    widths = spans[lidxs, 1] - spans[lidxs, 0]
    assert tmparr.shape == (128, S) and S == int(widths.sum()) and 40_000 < S < 60_000
    np.testing.assert_array_equal(tmpmap[:, 0], np.repeat(np.arange(len(lidxs)), widths))
    # structural validity (every output column is a resolved column of its source locus) on the first 300 loci
    n = 300
    sub_S = int(widths[:n].sum())
    R.check_replicate(seqarr, spans, lidxs[:n], tmparr[:, :sub_S], tmpmap[:sub_S])
    q = synth.random_quartets(128, 2500, seed=55)
    for sub in (True, False):
        rstat, rscor, flags = engine.resolve(q, sub)
        _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
        assert (flags == 0).all() and (o["flags"] == 0).all()
        np.testing.assert_array_equal(rstat, o_rstat)
        assert_close(rscor, o_rscor, o["svds"].max(axis=(1, 2))[:, None], "scores")


# ---------------------------------------------------------------------------------------------------
# drop-in surface
# ---------------------------------------------------------------------------------------------------
def test_infer_resolved_quartets_as_the_distributor_calls_it(tmp_path):
    """run_inference.py:219-220: `infer_resolved_quartets(database_file, nsamples, chunk, subsample_snps)` with
    `chunk` an islice over combinations -> the reference's own triple for c1, chunk by chunk."""
    from tetrad_amd import resolve_quartets as RQ
    g = load_golden("c1_T16_S5000")
    db = tmp_path / "snps.npz"
    np.savez(db, tmparr=g["tmparr"], tmpmap=g["tmpmap"])
    for sub, tag in ((True, "sub"), (False, "full")):
        rows = []
        for start in range(0, 1820, 455):                       # get_chunksize(1820, 4)
            chunk = islice(combinations(range(16), 4), start, start + 455)     # combinations.py:40-55
            rq, rstat, rscor = RQ.infer_resolved_quartets(db, 16, chunk, sub)
            assert rq.dtype == np.uint32 and rstat.dtype == np.uint32 and rscor.dtype == np.float64
            assert rq.shape == (455, 4) and rstat.shape == (455, 2) and rscor.shape == (455, 3)
            rows.append((rq, rstat, rscor))
        rq, rstat, rscor = (np.concatenate([r[i] for r in rows]) for i in range(3))
        np.testing.assert_array_equal(rq, g["quartets"])
        np.testing.assert_array_equal(rstat, g[f"{tag}_rstat"])
        assert_close(rscor, g[f"{tag}_rscor"], np.abs(g[f"{tag}_rscor"]).max(), "scores")


def test_new_infer_resolved_quartets_residency_is_safe(oracle):
    """The mirror uploads once per replicate -- and again whenever what it uploaded is no longer what the
    device holds: (1) somebody replaced the replicate through the shared engine, (2) the array was
    refilled in place, (3) a different array with the same id/shape came along."""
    from tetrad_amd import resolve_quartets as RQ
    gA, gB = load_golden("tree_T12_S2000"), load_golden("dense_T8_S400")
    A, mapA, qA = gA["tmparr"].copy(), gA["tmpmap"].copy(), gA["quartets"][:60]
    eng = RQ.get_engine(0)
    q, rstat, rscor = RQ.new_infer_resolved_quartets(A, mapA, qA, True)
    np.testing.assert_array_equal(rstat, gA["sub_rstat"][:60])
    gen = eng.data_generation
    RQ.new_infer_resolved_quartets(A, mapA, qA, False)
    assert eng.data_generation == gen, "second chunk of the same replicate must not upload again"
    # (1) replicate replaced behind the mirror's back
    eng.set_data(gB["tmparr"], gB["tmpmap"])
    _, rstat, _ = RQ.new_infer_resolved_quartets(A, mapA, qA, True)
    np.testing.assert_array_equal(rstat, gA["sub_rstat"][:60])
    # (2) refilled in place: same id, same shape, new content
    rng = np.random.default_rng(1)
    A[:] = rng.permutation(A.T).T
    _, rstat, rscor = RQ.new_infer_resolved_quartets(A, mapA, qA, True)
    _, o_rstat, o_rscor = oracle.new_infer_resolved_quartets(A, mapA, qA, True)
    np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
    assert_close(rscor, o_rscor, np.abs(o_rscor).max(), "scores after in-place refill")
    # an explicit replicate token skips the content fingerprint: same token = same replicate, by contract
    gen = eng.data_generation
    RQ.new_infer_resolved_quartets(A, mapA, qA, True, replicate_token=("rep", 3))
    RQ.new_infer_resolved_quartets(A, mapA, qA, True, replicate_token=("rep", 3))
    assert eng.data_generation == gen + 1
    RQ.invalidate()
    RQ.new_infer_resolved_quartets(A, mapA, qA, True, replicate_token=("rep", 3))
    assert eng.data_generation == gen + 2


@pytest.mark.parametrize("case", ["dense_T8_S400", "carry_T6_S2500", "edge_T7_S130"])
def test_kernel_level_mirrors(case):
    """subsample_chunk_to_matrices / full_chunk_to_matrices(seqs, locus, mask) exactly as
    resolve_quartets.py:212-223 calls them -> the reference's u32[3,16,16]."""
    from tetrad_amd import resolve_quartets as RQ
    g = load_golden(case)
    tmparr, tmpmap = g["tmparr"], g["tmpmap"]
    for qi in (0, len(g["quartets"]) // 2, len(g["quartets"]) - 1):
        sidx = g["quartets"][qi]
        seqs = tmparr[sidx, :]                                   # :212
        nmask0 = np.sum(seqs >= 78, axis=0)                      # :216
        nmask1 = np.sum(seqs == seqs[0], axis=0) == 4            # :218
        mask = nmask0 + nmask1                                   # :221
        got = RQ.subsample_chunk_to_matrices(seqs, tmpmap[:, 0], mask)
        assert got.dtype == np.uint32 and got.shape == (3, 16, 16)
        np.testing.assert_array_equal(got, g["sub_cmats"][qi])
        np.testing.assert_array_equal(RQ.full_chunk_to_matrices(seqs, tmpmap[:, 0], mask), g["full_cmats"][qi])


@pytest.mark.parametrize("cfg,nq", [("c3", 40_000), ("c2", 60_000)])
def test_scan_kernel_variants_agree_on_sorted_batches(cfg, nq):
    """Batches large enough to be sorted by (a,b,c) and scanned by the cooperative kernels with shared rows
    (>= 32 768 quartets): every alternative form of the scan -- the nibble-code kernel of rounds 1-3 with its lane-contiguous
    pattern park, two quartets per wavefront, row c through LDS, 8-wave workgroups, EXEC-masked counting in subsample mode and
    the walk in full mode, bank-private counters (scan_pb.hpp), the plane-record scan in full mode, no joint histogram -- must
    give bitwise the rows of the default forms (subsample: scan_f4.hpp, full: scan_dp.hpp), in both modes."""
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    T, S, _ = synth.CONFIGS[cfg]
    tmparr, tmpmap = synth.simulate_tmparr(T, S, synth.CONFIG_SEEDS[cfg])
    q = synth.all_quartets(T)[:nq] if cfg == "c2" else synth.random_quartets(T, nq, seed=5)
    defaults = {"park_t": 1, "scan_pair": 0, "share_c": 0, "scan_wg": 0, "scan_method": -1, "scan_dp": 1, "scan_f4": -1}
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        for sub in (True, False):
            base = eng.resolve(q, sub)
            for opts in ({"park_t": 0}, {"scan_pair": 1}, {"scan_pair": 1, "scan_method": 1 - int(sub)}, {"share_c": 1},
                         {"scan_wg": 8}, {"scan_method": 1 - int(sub)}, {"scan_method": 6}, {"scan_method": int(sub)},
                         {"scan_dp": 0}, {"scan_f4": 1}, {"scan_f4": 0}, {"scan_f4": 0, "scan_dp": 0}, {"scan_wg": 8, "scan_f4": 0}):
                for k, v in opts.items():
                    eng.set_option(k, v)
                got = eng.resolve(q, sub)
                for k in opts:
                    eng.set_option(k, defaults[k])
                for a, b in zip(base, got):
                    np.testing.assert_array_equal(a, b, err_msg=f"{opts} sub={sub}")


@pytest.mark.parametrize("case", ["dense_T8_S400", "tree_T12_S2000", "sparse_T10_S257", "edge_T7_S130", "carry_T6_S2500",
                                  "minrank_T14_S600"])
def test_row_f4_kernel_on_golden_cases(case):
    """SURVEY 8 row f4 as written -- the cooperative scan on ONE 12-byte plane record per taxon and 32-site lane-step, pattern
    bits of counted sites pulled out of the plane words inside the walk (scan_f4.hpp, option scan_f4) -- against the
    reference's count matrices and rows, both modes."""
    from test_gpu_parity import check_against
    from tetrad_amd.engine import QuartetEngine
    g = dict(load_golden(case))
    nq = len(g["quartets"])
    if nq < 64:                                   # the cooperative kernels take batches of 64 quartets or more: repeat the case
        reps = -(-64 // nq)
        for k, v in list(g.items()):
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == nq and k not in ("tmparr", "tmpmap"):
                g[k] = np.concatenate([v] * reps, axis=0)
    with QuartetEngine(0) as eng:
        eng.set_option("scan_f4", 1)
        eng.set_option("wg_min_quartets", 64)
        eng.set_data(g["tmparr"], g["tmpmap"])
        for mode in ("full", "sub"):
            rstat, rscor, flags, dbg = eng.resolve(g["quartets"], mode == "sub", debug=True)
            check_against(g, mode, rstat, rscor, flags, dbg)


def test_joint_histogram_scan_on_messy_batches(oracle):
    """scan_dp.hpp (full mode: two quartets that share (a,b,c) per wave, one joint histogram, folded at the end) against
    the one-quartet kernels and the oracle on what a caller may hand over: repeated quartets (runs of equal keys of any
    length), taxa in any order, taxon indices out of range, an all-missing taxon, quartets that arrive sorted or not,
    ordering switched off (then the kernel must not be chosen: its pairing relies on sorted keys)."""
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    rng = np.random.default_rng(17)
    T, S = 21, 4500
    tmparr, tmpmap = synth.simulate_tmparr(T, S, 99, p=0.08, missing=0.3)
    tmparr[5] = 78
    q = np.stack([rng.permutation(T)[:4] for _ in range(6000)]).astype(np.uint32)         # unsorted taxa, many repeats of (a,b,c)
    q[rng.integers(0, len(q), 300)] = q[0]                                                # one long run of equal quartets
    q[10:20] = np.sort(q[10:20], axis=1)
    bad = q.copy()
    bad[7, 3] = T + 3
    bad[8, 0] = 4_000_000_000
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        eng.set_option("dp_min_quartets", 2)
        for quartets in (q, q[np.lexsort((q[:, 3], q[:, 2], q[:, 1], q[:, 0]))], q[:1], q[:2], q[:5]):
            eng.set_option("scan_dp", 1)
            got = eng.resolve(quartets, False, debug=True)
            eng.set_option("scan_dp", 0)
            want = eng.resolve(quartets, False, debug=True)
            np.testing.assert_array_equal(got[3]["cmats"], want[3]["cmats"])
            for a, b in zip(got[:3], want[:3]):
                np.testing.assert_array_equal(a, b)
        _, o_rstat, _, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q[:400], False, debug=True)
        eng.set_option("scan_dp", 1)
        got = eng.resolve(q[:400], False, debug=True)
        np.testing.assert_array_equal(got[3]["cmats"], o["cmats"])
        np.testing.assert_array_equal(got[0][:, 1], o_rstat[:, 1])
        # taxon indices out of range (the device API does not look at the indices on the host): those rows are flagged
        # TQ_FLAG_BAD_INDEX by both kernels, the other rows are what they are without them
        import torch
        dev = torch.device("cuda:0")
        dq = torch.from_numpy(bad.astype(np.int64)).to(dev).to(torch.int32)               # (values >= 2^31 wrap: same bits)
        outs = []
        for dp in (1, 0):
            eng.set_option("scan_dp", dp)
            drs = torch.zeros((len(bad), 2), dtype=torch.int32, device=dev)
            dsc = torch.zeros((len(bad), 3), dtype=torch.float64, device=dev)
            dfl = torch.zeros(len(bad), dtype=torch.uint8, device=dev)
            eng.resolve_dev(dq.data_ptr(), len(bad), False, drs.data_ptr(), dsc.data_ptr(), dfl.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs.append((drs.cpu().numpy(), dsc.cpu().numpy(), dfl.cpu().numpy()))
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b)
        assert (outs[0][2][[7, 8]] & 4).all() and (np.delete(outs[0][2], [7, 8]) & 4).sum() == 0
        good = np.delete(np.arange(len(bad)), [7, 8])
        np.testing.assert_array_equal(outs[0][0][good].astype(np.uint32), eng.resolve(q, False)[0][good])
        # ordering off: natural order, no pairing, same rows
        eng.set_option("scan_dp", 1)
        eng.set_option("order", 0)
        g2 = eng.resolve(q, False)
        eng.set_option("order", 1)
        g3 = eng.resolve(q, False)
        for a, b in zip(g2, g3):
            np.testing.assert_array_equal(a, b)


def test_kernel_level_mirrors_honour_an_arbitrary_caller_mask(oracle):
    """The reference's count kernels count EVERY site their caller's mask leaves open (resolve_quartets.py:59-64,
    :89-95), invariant sites included -- only the worker's own mask (:216-218) always closes those.  A mask that is
    not the worker's: missing sites closed (the reference would index out of bounds on them), a random third of the
    others closed, invariant sites left open.  Against the oracle's restatement of the two kernels."""
    from tetrad_amd import resolve_quartets as RQ
    g = load_golden("tree_T12_S2000")
    tmparr, tmpmap = g["tmparr"], g["tmpmap"]
    rng = np.random.default_rng(3)
    for qi in (0, 200, 494):
        seqs = tmparr[g["quartets"][qi], :]
        missing = (seqs >= 78).any(axis=0)
        mask = (missing | (rng.random(seqs.shape[1]) < 0.33)).astype(np.int64)
        invariant_open = ((seqs == seqs[0]).all(axis=0) & (mask == 0)).sum()
        assert invariant_open > 100                                   # the case the worker's mask never produces
        for sub, fn, ofn in ((True, RQ.subsample_chunk_to_matrices, oracle.subsample_chunk_to_matrices),
                             (False, RQ.full_chunk_to_matrices, oracle.full_chunk_to_matrices)):
            want = ofn(seqs, tmpmap[:, 0], mask)
            got = fn(seqs, tmpmap[:, 0], mask)
            np.testing.assert_array_equal(got, want)
            if not sub:
                assert got[0].sum() == (mask == 0).sum()              # every open site counted once


def test_end_to_end_engine_rows_to_supertree(engine):
    """Data simulated on a known tree -> engine (all quartets) -> wQMC lines -> clean-room QMC: the generating tree
    comes back, for 16 taxa (c1) and for 40 taxa with a sampled quartet set."""
    from tetrad_amd import qmc, synth
    from test_qmc_tree import _bipartitions_from_children, _bipartitions_from_newick
    for T, S, seed, nq in ((16, 5000, synth.CONFIG_SEEDS["c1"], None), (40, 20000, 77, 40_000)):
        tmparr, tmpmap = synth.simulate_tmparr(T, S, seed)
        children, root = synth.random_tree_children(T, np.random.default_rng(seed))
        q = synth.all_quartets(T) if nq is None else synth.random_quartets(T, nq, seed=3)
        engine.set_data(tmparr, tmpmap)
        rstat, rscor, flags = engine.resolve(q, True)
        nwk = qmc.infer_supertree_from_arrays(q, rscor, rstat, T, weights=1)
        truth, got = _bipartitions_from_children(children, root, T), _bipartitions_from_newick(nwk, T)
        assert len(got & truth) >= len(truth) - (0 if T == 16 else 2), (len(got & truth), len(truth))


def test_diagnostic_modes_mark_every_row_and_refuse_unflagged_outputs():
    """Timing-diagnostic modes (scan_method 2..5, phases 1 / 2) produce wrong rows.  None may leave the library
    unmarked: every row carries TQ_FLAG_INVALID_DIAGNOSTIC, a device-API call without a flags array fails, the Python
    mirror of the reference's worker raises, and resetting the option gives clean rows again."""
    import torch
    from tetrad_amd import _lib, synth
    from tetrad_amd import resolve_quartets as RQ
    from tetrad_amd.engine import QuartetEngine
    T, S = 24, 6000
    tmparr, tmpmap = synth.simulate_tmparr(T, S, 5)
    q = synth.random_quartets(T, 4000, seed=2)
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        clean = eng.resolve(q, True)
        assert not (clean[2] & _lib.FLAG_INVALID_DIAGNOSTIC).any()
        for name, value, default in (("scan_method", 2, -1), ("scan_method", 4, -1), ("scan_method", 5, -1),
                                     ("phases", 1, 3), ("phases", 2, 3)):
            eng.set_option(name, value)
            _, _, flags = eng.resolve(q, True)
            assert (flags & _lib.FLAG_INVALID_DIAGNOSTIC).all(), (name, value)
            d_q = torch.from_numpy(q.view(np.int32)).cuda()
            d_rstat = torch.zeros((len(q), 2), dtype=torch.int32, device="cuda")
            d_rscor = torch.zeros((len(q), 3), dtype=torch.float64, device="cuda")
            with pytest.raises(_lib.TetradHipError, match="diagnostic"):
                eng.resolve_dev(d_q.data_ptr(), len(q), True, d_rstat.data_ptr(), d_rscor.data_ptr(), 0, 0)
            torch.cuda.synchronize()
            eng.set_option(name, default)
        again = eng.resolve(q, True)
        for a, b in zip(clean, again):
            np.testing.assert_array_equal(a, b)
    # the drop-in mirror (tetrad_amd.resolve_quartets) refuses such rows
    eng = RQ.get_engine(0)
    try:
        eng.set_option("scan_method", 4)
        with pytest.raises(RuntimeError, match="diagnostic"):
            RQ.new_infer_resolved_quartets(tmparr, tmpmap, q[:2000], True)
    finally:
        eng.set_option("scan_method", -1)
    RQ.new_infer_resolved_quartets(tmparr, tmpmap, q[:2000], True)


def test_batch_option_is_clamped():
    """`batch` reaches hipCUB as int: values beyond 2^31 - 1 are clamped, not wrapped."""
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    tmparr, tmpmap = synth.simulate_tmparr(12, 3000, 11)
    q = synth.all_quartets(12)
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        base = eng.resolve(q, True)
        eng.set_option("batch", 1 << 40)
        got = eng.resolve(q, True)
        for a, b in zip(base, got):
            np.testing.assert_array_equal(a, b)


def test_device_api_calls_on_different_streams_and_host_api_are_ordered(oracle):
    """The *_dev entry points of one context share its scratch.  A resolve enqueued on a side stream BEHIND a long-running
    kernel, a second one on another stream right after it, then a host-API call with different quartets straight away:
    every result must equal the oracle's (the library orders the calls in call order, include/tetrad_hip.h)."""
    import torch
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    T, S = 20, 9000
    tmparr, tmpmap = synth.simulate_tmparr(T, S, 8)
    qa = synth.random_quartets(T, 3000, seed=1)
    qb = synth.random_quartets(T, 3000, seed=2)
    qc = synth.random_quartets(T, 3000, seed=3)
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        dev = {}
        for name, q in (("a", qa), ("b", qb)):
            dev[name] = (torch.from_numpy(q.view(np.int32)).cuda(), torch.zeros((len(q), 2), dtype=torch.int32, device="cuda"),
                         torch.zeros((len(q), 3), dtype=torch.float64, device="cuda"), torch.zeros(len(q), dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        with torch.cuda.stream(s1):
            for _ in range(20):                      # tens of milliseconds of work in front of the first resolve
                big.mul_(1.0001)
        d = dev["a"]
        eng.resolve_dev(d[0].data_ptr(), len(qa), True, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), s1.cuda_stream)
        d = dev["b"]
        eng.resolve_dev(d[0].data_ptr(), len(qb), False, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), s2.cuda_stream)
        host = eng.resolve(qc, True)                 # host API: waits for both
        torch.cuda.synchronize()
        for (q, sub, rstat, rscor) in ((qa, True, dev["a"][1].cpu().numpy().view(np.uint32), dev["a"][2].cpu().numpy()),
                                       (qb, False, dev["b"][1].cpu().numpy().view(np.uint32), dev["b"][2].cpu().numpy()),
                                       (qc, True, host[0], host[1])):
            _, o_rstat, o_rscor = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, sub)
            np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
            np.testing.assert_allclose(rscor, o_rscor, rtol=1e-6, atol=1e-9)


def test_debug_bdsqr_hook_reproduces_the_pipeline_values():
    """`tq_debug_bdsqr` (tools/bdsqr_order.py measures wave orderings with it) runs the QR kernel alone on bidiagonals
    given on the host: on the bidiagonals the pipeline itself left behind it must return the pipeline's own singular
    values, bit for bit, in the given order and in a permuted one (a lane's arithmetic does not depend on its wave-mates),
    together with plausible work counters."""
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    tmparr, tmpmap = synth.simulate_tmparr(20, 6000, 13)
    q = synth.random_quartets(20, 3000, seed=4)
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        eng.set_option("svd_streams", 1)
        eng.set_option("svd_chunk", len(q))
        eng.resolve(q, True)
        de = eng.debug_fetch("de", len(q))
        sv_pipe = eng.debug_fetch("sv", len(q))
        sv, steps, sweeps, ms = eng.debug_bdsqr(de, reps=2)
        np.testing.assert_array_equal(sv, sv_pipe)
        assert ms > 0 and steps.min() >= 15 and steps.max() < 2000 and sweeps.min() >= 1
        perm = np.random.default_rng(0).permutation(len(de))
        sv_p, steps_p, _, _ = eng.debug_bdsqr(de[perm], reps=1)
        np.testing.assert_array_equal(sv_p, sv[perm])
        np.testing.assert_array_equal(steps_p, steps[perm])
