"""Clean-room weighted Quartet MaxCut (tq_qmc_tree, host C++).  PARITY UNPINNED: the reference delegates this step
to a prebuilt binary without source (run_inference.py:146-166), holds no expected tree for any quartet set, and the
binary may not be run.  Tested here: what any correct implementation must do -- recover the generating tree from
its own quartets (complete, sampled, noisy + weighted), determinism, malformed input."""
from itertools import combinations

import numpy as np
import pytest

from tetrad_amd import qmc, synth


def _tree_dist(children, root, T):
    """pairwise path lengths (unit edges) between the tips of a rooted binary tree given as children[node]."""
    parent = {}
    for p, (a, b) in children.items():
        parent[a] = p
        parent[b] = p
    def path(t):
        out = [t]
        while out[-1] in parent:
            out.append(parent[out[-1]])
        return out
    paths = [path(t) for t in range(T)]
    D = np.zeros((T, T), int)
    for i in range(T):
        pi = {n: k for k, n in enumerate(paths[i])}
        for j in range(i + 1, T):
            for k, n in enumerate(paths[j]):
                if n in pi:
                    D[i, j] = D[j, i] = k + pi[n]
                    break
    return D


def _true_splits(D, quartets):
    """the split of each quartet on the tree: the pairing with the smallest sum of path lengths (four-point condition)"""
    out = np.empty_like(quartets)
    for i, (a, b, c, d) in enumerate(quartets):
        s = [D[a, b] + D[c, d], D[a, c] + D[b, d], D[a, d] + D[b, c]]
        k = int(np.argmin(s))
        out[i] = [(a, b, c, d), (a, c, b, d), (a, d, b, c)][k]
    return out


def _bipartitions_from_children(children, root, T):
    below = {}
    def rec(n):
        if n < T:
            below[n] = frozenset([n])
        else:
            a, b = children[n]
            below[n] = rec(a) | rec(b)
        return below[n]
    rec(root)
    allt = frozenset(range(T))
    return {min(s, allt - s, key=lambda x: (len(x), sorted(x))) for s in below.values() if 1 < len(s) < T - 1}


def _bipartitions_from_newick(nwk, T):
    assert nwk.endswith(";")
    pos = 0
    splits = []
    def parse():
        nonlocal pos
        if nwk[pos] == "(":
            pos += 1
            s = frozenset()
            while True:
                s |= parse()
                if nwk[pos] == ",":
                    pos += 1
                    continue
                assert nwk[pos] == ")"
                pos += 1
                break
            splits.append(s)
            return s
        j = pos
        while nwk[j].isdigit():
            j += 1
        t = int(nwk[pos:j])
        pos = j
        return frozenset([t])
    top = parse()
    assert nwk[pos] == ";" and top == frozenset(range(T)), "every taxon exactly once"
    allt = frozenset(range(T))
    return {min(s, allt - s, key=lambda x: (len(x), sorted(x))) for s in splits if 1 < len(s) < T - 1}


@pytest.mark.parametrize("T,seed", [(5, 1), (8, 2), (13, 3), (24, 4), (40, 5)])
def test_recovers_the_generating_tree_from_all_its_quartets(T, seed):
    rng = np.random.default_rng(seed)
    children, root = synth.random_tree_children(T, rng)
    D = _tree_dist(children, root, T)
    quartets = np.array(list(combinations(range(T), 4)), np.uint32)
    splits = _true_splits(D, quartets)
    nwk = qmc.qmc_tree(splits, None, T, seed=7)
    assert _bipartitions_from_newick(nwk, T) == _bipartitions_from_children(children, root, T)
    assert qmc.qmc_tree(splits, None, T, seed=7) == nwk                     # deterministic in the seed
    # row order does not matter (the reference has to shuffle its input file, run_inference.py:323-327)
    perm = rng.permutation(len(splits))
    assert _bipartitions_from_newick(qmc.qmc_tree(splits[perm], None, T, seed=7), T) == _bipartitions_from_newick(nwk, T)


def test_recovers_the_tree_from_a_random_sample_of_quartets():
    """the reference's recommended sampling: >= nsamples**2.8 quartets (write_database.py:85)"""
    T = 30
    rng = np.random.default_rng(11)
    children, root = synth.random_tree_children(T, rng)
    D = _tree_dist(children, root, T)
    quartets = synth.random_quartets(T, int(T ** 2.8), seed=3)
    nwk = qmc.qmc_tree(_true_splits(D, quartets), None, T, seed=1)
    assert _bipartitions_from_newick(nwk, T) == _bipartitions_from_children(children, root, T)


def test_weights_outvote_noise():
    """20 % of the quartets wrong but light, the correct ones heavy: weighted -> the true tree; and the unweighted run
    on the same rows still finds most of it."""
    T = 16
    rng = np.random.default_rng(21)
    children, root = synth.random_tree_children(T, rng)
    D = _tree_dist(children, root, T)
    quartets = np.array(list(combinations(range(T), 4)), np.uint32)
    splits = _true_splits(D, quartets)
    wrong = rng.random(len(splits)) < 0.2
    noisy = splits.copy()
    noisy[wrong] = noisy[wrong][:, [0, 2, 1, 3]]                            # a,c|b,d instead of a,b|c,d
    w = np.where(wrong, 0.05, 1.0)
    truth = _bipartitions_from_children(children, root, T)
    assert _bipartitions_from_newick(qmc.qmc_tree(noisy, w, T), T) == truth
    got = _bipartitions_from_newick(qmc.qmc_tree(noisy, None, T), T)
    assert len(got & truth) >= len(truth) - 2


def test_degenerate_inputs():
    assert qmc.qmc_tree(np.zeros((0, 4), np.uint32), None, 1) == "0;"
    assert _bipartitions_from_newick(qmc.qmc_tree(np.zeros((0, 4), np.uint32), None, 6), 6) == set()    # a star
    one = np.array([[0, 1, 2, 3]], np.uint32)
    assert _bipartitions_from_newick(qmc.qmc_tree(one, None, 4), 4) == {frozenset([0, 1])}
    # taxa no quartet mentions still appear exactly once; repeated taxa inside a row are ignored
    rows = np.array([[0, 1, 2, 3], [0, 1, 2, 2], [0, 1, 2, 3]], np.uint32)
    bips = _bipartitions_from_newick(qmc.qmc_tree(rows, np.array([1.0, 5.0, 1.0]), 7), 7)
    allt = frozenset(range(7))
    # the one informative quartet 0,1|2,3 is displayed: some edge has 0,1 on one side and 2,3 on the other
    assert any(({0, 1} <= s and not ({2, 3} & s)) or ({0, 1} <= allt - s and not ({2, 3} & (allt - s))) for s in bips)
    from tetrad_amd._lib import TetradHipError
    with pytest.raises(TetradHipError):
        qmc.qmc_tree(np.array([[0, 1, 2, 9]], np.uint32), None, 4)         # taxon >= ntaxa


def test_file_mirror_and_array_entry(tmp_path):
    """run_qmc(in, out, use_weights) on a wQMC file; infer_supertree_from_arrays straight from a result triple."""
    T = 10
    rng = np.random.default_rng(5)
    children, root = synth.random_tree_children(T, rng)
    D = _tree_dist(children, root, T)
    quartets = np.array(list(combinations(range(T), 4)), np.uint32)
    splits = _true_splits(D, quartets)
    f = tmp_path / "qmc_in.txt"
    f.write_text("".join("%d,%d|%d,%d:%.5f\n" % (*r, 1.0) for r in splits.tolist()))
    qmc.run_qmc(f, tmp_path / "out.nwk", True, ntaxa=T)
    truth = _bipartitions_from_children(children, root, T)
    assert _bipartitions_from_newick((tmp_path / "out.nwk").read_text().strip(), T) == truth
    # a result triple whose topology column encodes the true split of each (sorted) quartet
    topo = np.array([[tuple(s) == (a, b, c, d), tuple(s) == (a, c, b, d), tuple(s) == (a, d, b, c)].index(True)
                     for s, (a, b, c, d) in zip(splits.tolist(), quartets.tolist())], np.uint32)
    rstat = np.stack([topo, np.full(len(topo), 100, np.uint32)], axis=1)
    rscor = np.full((len(topo), 3), 5.0)
    rscor[np.arange(len(topo)), topo] = 1.0
    for weights in (0, 1, 2, 3):
        nwk = qmc.infer_supertree_from_arrays(quartets, rscor, rstat, T, weights=weights)
        assert _bipartitions_from_newick(nwk, T) == truth


def test_relabel_tree():
    nwk = "((0,1),(2,3),(4,10));"
    names = {0: "a", 1: "b b", 2: "c", 3: "d", 4: "e", 10: "taxon_10"}
    assert qmc.relabel_tree(nwk, names) == "((a,'b b'),(c,d),(e,taxon_10));"
    assert qmc.relabel_tree("(0,1,2);", ["x", "y", "z"]) == "(x,y,z);"
    with pytest.raises(KeyError):
        qmc.relabel_tree("(0,1,7);", ["x", "y", "z"])


@pytest.mark.parametrize("mode", ["sub", "full"])
@pytest.mark.parametrize("weights", [0, 1, 2, 3])
def test_end_to_end_from_the_reference_rows_of_c1(mode, weights):
    """BASELINE.json configs[0] end to end: the REFERENCE's own rows for all 1 820 quartets of the c1 data
    (tests/golden/c1_T16_S5000.npz) -> wQMC lines (tq_format_qmc) -> tq_qmc_tree recovers the tree the data were
    simulated on, for every weight strategy of run_inference.py:280-297."""
    from conftest import load_golden
    g = load_golden("c1_T16_S5000")
    children, root = synth.random_tree_children(16, np.random.default_rng(synth.CONFIG_SEEDS["c1"]))   # simulate_tmparr's tree
    nwk = qmc.infer_supertree_from_arrays(g["quartets"], g[f"{mode}_rscor"], g[f"{mode}_rstat"], 16, weights=weights)
    assert _bipartitions_from_newick(nwk, 16) == _bipartitions_from_children(children, root, 16)


@pytest.mark.parametrize("weights", [0, 1, 2, 3])
@pytest.mark.parametrize("min_snps,min_ratio", [(0, 1.0), (500, 1.3)])
def test_qmc_splits_equal_the_parsed_lines(weights, min_snps, min_ratio):
    """tq_qmc_splits (arrays) == the lines of tq_format_qmc parsed back, bit for bit, incl. the "%.5f" weights"""
    from tetrad_amd import qmc_format
    rng = np.random.default_rng(weights)
    n = 4000
    q = np.sort(rng.integers(0, 60, size=(n, 4)), axis=1).astype(np.uint32)
    sc = rng.gamma(2.0, 20.0, size=(n, 3))
    sc[::37] = 0.001
    sc[5::101, 0] = 0.0
    st = np.stack([rng.integers(0, 3, size=n), rng.integers(0, 3000, size=n)], axis=1).astype(np.uint32)
    sp, w = qmc.qmc_splits(q, sc, st, weights, min_snps, min_ratio)
    sp2, w2 = qmc.parse_qmc_lines(qmc_format.qmc_lines(q, sc, st, weights, min_snps, min_ratio))
    np.testing.assert_array_equal(sp, sp2)
    np.testing.assert_array_equal(w, w2)


def test_infer_supertree_file_pipeline(tmp_path):
    """quartets TSV (as the distributor writes it) -> wQMC input file -> tree file -> named newick"""
    from conftest import load_golden
    from tetrad_amd import distributor
    g = load_golden("c1_T16_S5000")
    tsv = tmp_path / "x.quartets_0.tsv"
    tsv.write_bytes(distributor.format_tsv_bytes(g["quartets"], g["sub_rscor"], g["sub_rstat"]))
    names = {i: f"sample_{i}" for i in range(16)}
    nwk = qmc.infer_supertree(tsv, tmp_path / "qmc_in.txt", tmp_path / "qmc_out.txt", 16, weights=1, samples=names)
    assert nwk.count("sample_") == 16 and nwk.endswith(";")
    numeric = (tmp_path / "qmc_out.txt").read_text().strip()
    children, root = synth.random_tree_children(16, np.random.default_rng(synth.CONFIG_SEEDS["c1"]))
    assert _bipartitions_from_newick(numeric, 16) == _bipartitions_from_children(children, root, 16)
    assert len((tmp_path / "qmc_in.txt").read_text().splitlines()) == 1820
