"""GPU: bootstrap replicates built on the device (tq_set_source / tq_bootstrap / tq_get_data).

RNG-stream parity with the reference is unpinned (numba's stream), so the device
replicate is checked against the structural definition (oracle/resample.check_replicate), for
determinism, for the distribution of the random choices, and end to end: resolving quartets on the
device-built replicate must equal the oracle run on the exported (tmparr, tmpmap)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from tetrad_amd.engine import QuartetEngine
    eng = QuartetEngine(0)
    yield eng
    eng.close()


def test_device_replicate_is_structurally_valid(engine):
    from oracle import resample as R
    g = load_golden("resample_T7_S300")
    engine.set_source(g["seqarr"], g["spans"])
    S = engine.bootstrap(g["lidxs"], 111, 222)
    tmparr, tmpmap = engine.get_data()
    assert tmparr.shape[1] == S == int((g["spans"][g["lidxs"], 1] - g["spans"][g["lidxs"], 0]).sum())
    R.check_replicate(g["seqarr"], g["spans"], g["lidxs"], tmparr, tmpmap)
    # deterministic in the seeds, different for different seeds
    engine.bootstrap(g["lidxs"], 111, 222)
    again = engine.get_data()
    np.testing.assert_array_equal(again[0], tmparr)
    engine.bootstrap(g["lidxs"], 112, 222)
    assert not np.array_equal(engine.get_data()[0], tmparr)


def test_random_choices_are_unbiased(engine):
    """Ambiguity codes resolve ~50/50 to their two bases; within-locus shuffles are ~uniform."""
    rng = np.random.default_rng(3)
    T, nloci, w = 4, 3000, 3
    S0 = nloci * w
    seqarr = np.full((T, S0), 65, np.uint8)
    seqarr[0] = np.tile(np.array([65, 67, 71], np.uint8), nloci)      # column identity inside each locus
    seqarr[1] = 82                                                     # R -> G(2) / A(0)
    seqarr[2] = 89                                                     # Y -> T(3) / C(1)
    spans = np.stack([np.arange(nloci) * w, np.arange(nloci) * w + w], axis=1)
    engine.set_source(seqarr, spans)
    engine.bootstrap(np.arange(nloci), 7, 9)
    tmparr, tmpmap = engine.get_data()
    assert set(np.unique(tmparr[1])) == {0, 2} and set(np.unique(tmparr[2])) == {1, 3}
    assert abs((tmparr[1] == 2).mean() - 0.5) < 0.03 and abs((tmparr[2] == 3).mean() - 0.5) < 0.03
    perms = tmparr[0].reshape(nloci, w)
    keys, counts = np.unique(perms[:, 0] * 16 + perms[:, 1] * 4 + perms[:, 2], return_counts=True)
    assert len(keys) == 6 and counts.min() > nloci / 6 * 0.8 and counts.max() < nloci / 6 * 1.2


def test_resolve_on_device_replicate_matches_oracle(engine, oracle):
    """c5-shaped flow: source -> device bootstrap -> resolve, vs the oracle on the exported replicate."""
    from tetrad_amd import bootstrap, synth
    tmparr0, tmpmap0 = synth.simulate_tmparr(14, 6000, seed=8)
    ascii_ = np.array([65, 67, 71, 84], np.uint8)
    seqarr = np.where(tmparr0 <= 3, ascii_[np.minimum(tmparr0, 3)], 78).astype(np.uint8)
    rs = np.random.default_rng(0)
    amb = rs.random(seqarr.shape) < 0.02
    seqarr[amb] = rs.choice(np.array([82, 75, 83, 89, 87, 77], np.uint8), size=int(amb.sum()))
    spans = bootstrap.get_spans(tmpmap0)
    engine.set_source(seqarr, spans)
    rng = np.random.default_rng(2024)
    quartets = synth.all_quartets(14)
    for rep in range(3):
        S = bootstrap.resample_tmp_database(engine, rng)
        tmparr, tmpmap = engine.get_data()
        assert tmparr.shape == (14, S)
        for sub in (True, False):
            rstat, rscor, flags = engine.resolve(quartets, sub)
            _, o_rstat, o_rscor, o = oracle.new_infer_resolved_quartets(tmparr, tmpmap, quartets, sub, debug=True)
            np.testing.assert_array_equal(rstat[:, 1], o_rstat[:, 1])
            ok = ((flags | o["flags"]) & 3) == 0
            np.testing.assert_array_equal(rstat[ok, 0], o_rstat[ok, 0])
            smax = o["svds"].max(axis=(1, 2))[:, None]
            assert np.all(np.abs(rscor - o_rscor) <= 1e-6 * np.abs(o_rscor) + 1e-12 * smax)
    # the project Generator advanced exactly as the reference's would (3 draws per replicate)
    chk = np.random.default_rng(2024)
    for rep in range(3):
        chk.choice(len(spans), len(spans), replace=True); chk.integers(2**31); chk.integers(2**31)
    assert rng.bit_generator.state == chk.bit_generator.state


def test_bootstrap_errors(engine):
    from tetrad_amd.engine import QuartetEngine, TetradHipError
    with QuartetEngine(0) as eng:
        with pytest.raises(TetradHipError) as e:
            eng.bootstrap(np.zeros(3, np.int64), 1, 2)
        assert e.value.code == -4
        with pytest.raises(TetradHipError):
            eng.set_source(np.zeros((3, 10), np.uint8), np.array([[0, 5], [5, 11]]))   # span beyond S0
        eng.set_source(np.full((3, 10), 65, np.uint8), np.array([[0, 5], [5, 10]]))
        with pytest.raises(TetradHipError):
            eng.bootstrap(np.array([0, 2], np.int64), 1, 2)                              # locus index out of range
        with pytest.raises(TetradHipError):
            eng.bootstrap(np.array([0], np.int64), 1, 2)                                 # wrong count
