"""GPU: the bootstrap-replicate loop (tetrad_amd/replicates.py, mirror of run_inference.py:378-407) and the
opt-in device quartet sampler."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _small_source():
    from tetrad_amd import synth
    return synth.make_c5_source(T=14, S=6000, seed=8, ambiguous=0.02)


def test_device_sampler_draws_distinct_uniform_quartets():
    """tq_sample_quartets_dev: Q distinct ranks < C(T,4), quartets = their lexicographic unranking, uniform
    over the rank space and over taxa, deterministic in the seed, different for different seeds."""
    import torch
    from math import comb
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    tmparr, tmpmap = synth.simulate_tmparr(40, 300, seed=1)
    dev = torch.device("cuda:0")
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        N = comb(40, 4)                                             # 91 390
        for Q in (1, 1000, N // 3, N):                              # N: the whole space, a permutation
            d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
            d_r = torch.empty(Q, dtype=torch.int64, device=dev)
            eng.sample_quartets_dev(12345, Q, d_q.data_ptr(), d_r.data_ptr())
            torch.cuda.synchronize()
            r = d_r.cpu().numpy()
            assert r.min() >= 0 and r.max() < N and len(np.unique(r)) == Q
            np.testing.assert_array_equal(d_q.cpu().numpy().view(np.uint32), synth.unrank_quartets(r, 40))
        Q = N // 3
        d_q2 = torch.empty((Q, 4), dtype=torch.int32, device=dev)
        d_r2 = torch.empty(Q, dtype=torch.int64, device=dev)
        eng.sample_quartets_dev(12345, Q, d_q2.data_ptr(), d_r2.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_r2.cpu().numpy(), r[:Q])    # a prefix of the same permutation
        eng.sample_quartets_dev(12346, Q, d_q2.data_ptr(), d_r2.data_ptr())
        torch.cuda.synchronize()
        r2 = d_r2.cpu().numpy()
        assert len(np.intersect1d(r2, r[:Q])) < 0.45 * Q            # ~1/3 expected for independent samples
        # uniformity: 64 equal buckets of the rank space, chi-square against the hypergeometric-ish expectation
        counts = np.bincount((r2.astype(np.float64) / N * 64).astype(int), minlength=64)
        exp = Q / 64
        assert ((counts - exp) ** 2 / exp).sum() < 64 * 2.0
        # the order is random too: no monotone runs
        assert abs(np.corrcoef(np.arange(Q), r2)[0, 1]) < 0.02
        with pytest.raises(Exception):
            eng.sample_quartets_dev(1, N + 1, d_q2.data_ptr(), 0)


def test_device_sampler_matches_the_distribution_of_choice_without_replacement():
    """The opt-in device sampler is documented as distribution-equal to `rng.choice(C(T,4), Q, replace=False)`
    (combinations.py:109-114), not stream-equal.  Chi-square tests at the c3 shape (128 taxa, 1e5 of 10.7e6):
    ranks over 256 equal bins, first-taxon frequencies against their exact law C(T-a-1,3)/C(T,4), last-taxon
    frequencies likewise, for related seeds (s, s+1, s with a high bit flipped: the round keys come from a
    splitmix64 chain, so related seeds must give unrelated samples), and against NumPy's own sampler as a
    two-sample check."""
    import torch
    from math import comb
    from tetrad_amd import synth
    from tetrad_amd.engine import QuartetEngine
    T, Q = 128, 100_000
    N = comb(T, 4)
    tmparr, tmpmap = synth.simulate_tmparr(T, 64, seed=1)
    dev = torch.device("cuda:0")
    first_p = np.array([comb(T - a - 1, 3) for a in range(T)], float) / N
    last_p = np.array([comb(d, 3) for d in range(T)], float) / N
    keep_f, keep_l = first_p * Q >= 20, last_p * Q >= 20            # pool the thin tails out of the statistic

    def chi2(counts, p, keep):
        e = p[keep] * Q
        return float((((counts[keep] - e) ** 2) / e).sum()), int(keep.sum()) - 1

    def ok(stat, dof):                                              # within 5 sigma of a chi-square's mean
        return abs(stat - dof) < 5 * np.sqrt(2 * dof)

    samples = {}
    with QuartetEngine(0) as eng:
        eng.set_data(tmparr, tmpmap)
        for seed in (7, 8, 7 ^ (1 << 40), 0, 2**63 - 1):
            d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
            d_r = torch.empty(Q, dtype=torch.int64, device=dev)
            eng.sample_quartets_dev(seed, Q, d_q.data_ptr(), d_r.data_ptr())
            torch.cuda.synchronize()
            r, q = d_r.cpu().numpy(), d_q.cpu().numpy()
            assert len(np.unique(r)) == Q
            bins = np.bincount((r // (N // 256 + 1)).astype(int), minlength=256)
            width = np.diff(np.minimum(np.arange(257) * (N // 256 + 1), N)).astype(float)
            stat = float((((bins - Q * width / N) ** 2) / (Q * width / N)).sum())
            assert ok(stat, 255), ("rank bins", seed, stat)
            s1, d1 = chi2(np.bincount(q[:, 0], minlength=T), first_p, keep_f)
            s2, d2 = chi2(np.bincount(q[:, 3], minlength=T), last_p, keep_l)
            assert ok(s1, d1) and ok(s2, d2), ("taxon frequencies", seed, s1, d1, s2, d2)
            samples[seed] = r
    # related seeds: overlaps like independent samples (expected Q*Q/N = 937, sd ~ 30)
    for a, b in ((7, 8), (7, 7 ^ (1 << 40)), (0, 2**63 - 1)):
        inter = len(np.intersect1d(samples[a], samples[b]))
        assert abs(inter - Q * Q / N) < 6 * np.sqrt(Q * Q / N), (a, b, inter)
    # two-sample check against NumPy's sampler: the same bin statistic between the two samples
    ref = np.random.default_rng(5).choice(N, size=Q, replace=False)
    b1 = np.bincount((samples[7] // (N // 256 + 1)).astype(int), minlength=256).astype(float)
    b2 = np.bincount((ref // (N // 256 + 1)).astype(int), minlength=256).astype(float)
    stat = float((((b1 - b2) ** 2) / (b1 + b2)).sum())
    assert ok(stat, 255), ("two-sample", stat)


@pytest.mark.parametrize("sampler", ["host", "device"])
def test_replicate_runner_equals_step_by_step(oracle, sampler):
    """Four replicates through the pipelined runner == the same draws applied step by step with a second
    engine (synchronous bootstrap + resolve), and == the oracle on the exported replicate for a sample; the
    Generator ends where the reference's would."""
    import torch
    from math import comb
    from tetrad_amd import bootstrap, synth
    from tetrad_amd.engine import QuartetEngine
    from tetrad_amd.replicates import ReplicateRunner
    seqarr, maparr, spans = _small_source()
    Q, nreps, seed = 700, 4, 99
    got = {}
    with QuartetEngine(0) as eng:
        runner = ReplicateRunner(eng, seqarr, spans, Q, seed=seed, sampler=sampler, ahead=2)
        stats = runner.run(nreps, True, on_result=lambda k, S, a, b, c: got.__setitem__(k, (S, a.copy(), b.copy(), c.copy())))
        final_state = runner.rng.bit_generator.state
        runner.close()
    assert sorted(got) == list(range(nreps)) and len(stats["sites"]) == nreps
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    with QuartetEngine(0) as eng:
        eng.set_source(seqarr, spans)
        for k in range(nreps):
            lidxs, s1, s2 = bootstrap.draw_replicate(len(spans), rng)
            S = eng.bootstrap(lidxs, s1, s2)
            if sampler == "host":
                idx = rng.choice(comb(14, 4), size=Q, replace=False)
                q = synth.unrank_quartets(idx, 14)
            else:
                sd = int(rng.integers(2**63))
                d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
                eng.sample_quartets_dev(sd, Q, d_q.data_ptr(), 0)
                torch.cuda.synchronize()
                q = d_q.cpu().numpy().view(np.uint32)
                assert len(np.unique(q, axis=0)) == Q
            rstat, rscor, flags = eng.resolve(q, True)
            assert got[k][0] == S
            np.testing.assert_array_equal(got[k][1], rstat)
            np.testing.assert_array_equal(got[k][2], rscor)
            np.testing.assert_array_equal(got[k][3], flags)
            if k == nreps - 1:
                tmparr, tmpmap = eng.get_data()
                _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q[:200], True, debug=True)
                np.testing.assert_array_equal(rstat[:200, 1], o_rstat[:, 1])
                ok = ((flags[:200] | o["flags"]) & 3) == 0
                np.testing.assert_array_equal(rstat[:200][ok, 0], o_rstat[ok, 0])
    # the producer may have drawn ahead, but never beyond the replicates it was asked for
    assert final_state == rng.bit_generator.state
    assert runner.rng_state_after[-1] == rng.bit_generator.state


@pytest.mark.parametrize("sampler", ["host", "device"])
def test_bootstrap_trees_end_to_end(sampler):
    """run_inference.py:378-407 with the supertree step: replicates -> rows + sampled quartets on the host -> one tree per
    replicate.  Every tree must equal the tree built step by step from the same draws, hold every taxon once, and --
    the data being simulated on a known tree -- most replicates recover most of that tree."""
    import torch
    from math import comb
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).parent))
    from test_qmc_tree import _bipartitions_from_children, _bipartitions_from_newick
    from tetrad_amd import bootstrap, qmc, synth
    from tetrad_amd.engine import QuartetEngine
    from tetrad_amd.replicates import bootstrap_trees
    T, S, seed = 14, 6000, 8
    seqarr, maparr, spans = synth.make_c5_source(T=T, S=S, seed=seed, ambiguous=0.02)
    children, root = synth.random_tree_children(T, np.random.default_rng(seed))
    truth = _bipartitions_from_children(children, root, T)
    Q, nboots = 800, 4
    with QuartetEngine(0) as eng:
        trees = bootstrap_trees(eng, seqarr, spans, Q, nboots, weights=1, seed=21, sampler=sampler, workers=2)
    assert len(trees) == nboots
    rng = np.random.default_rng(21)
    dev = torch.device("cuda:0")
    agree = []
    with QuartetEngine(0) as eng:
        eng.set_source(seqarr, spans)
        for k in range(nboots):
            lidxs, s1, s2 = bootstrap.draw_replicate(len(spans), rng)
            eng.bootstrap(lidxs, s1, s2)
            if sampler == "host":
                q = synth.unrank_quartets(rng.choice(comb(T, 4), size=Q, replace=False), T)
            else:
                d_q = torch.empty((Q, 4), dtype=torch.int32, device=dev)
                eng.sample_quartets_dev(int(rng.integers(2**63)), Q, d_q.data_ptr(), 0)
                torch.cuda.synchronize()
                q = d_q.cpu().numpy().view(np.uint32)
            rstat, rscor, _ = eng.resolve(q, True)
            assert trees[k] == qmc.infer_supertree_from_arrays(q, rscor, rstat, T, 1, seed=k)
            got = _bipartitions_from_newick(trees[k], T)
            agree.append(len(got & truth))
    assert max(agree) >= len(truth) - 2, (agree, len(truth))
