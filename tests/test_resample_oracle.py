"""CPU: the bootstrap-producer restatement (oracle/resample.py) against the reference's own
functions (tests/golden/resample_*.npz, spans_toy.npz), and the host mirror of get_spans."""
import numpy as np

from conftest import load_golden
from oracle import resample as R
from tetrad_amd import bootstrap as B


def test_spans_match_reference():
    for name in ("resample_T7_S300", "spans_toy"):
        g = load_golden(name)
        np.testing.assert_array_equal(R.get_spans(g["maparr"]), g["spans"])
        np.testing.assert_array_equal(B.get_spans(g["maparr"]), g["spans"])


def test_resample_and_resolve_match_reference_logic():
    g = load_golden("resample_T7_S300")
    tmparr, tmpmap = R.resample(g["seqarr"], g["spans"], g["lidxs"], int(g["seed_resample"]))
    np.testing.assert_array_equal(tmparr, g["tmparr"])
    np.testing.assert_array_equal(tmpmap, g["tmpmap"])
    resolved = R.resolve_ambigs(tmparr.copy(), int(g["seed_ambig"]))
    np.testing.assert_array_equal(resolved, g["resolved"])
    # the reference's own output satisfies the structural checker used for the device bootstrap
    R.check_replicate(g["seqarr"], g["spans"], g["lidxs"], R.recode(resolved.copy()), tmpmap)


def test_checker_rejects_wrong_replicates():
    g = load_golden("resample_T7_S300")
    good = R.recode(g["resolved"].copy())
    bad = good.copy()
    k = int(np.argmax((g["spans"][g["lidxs"], 1] - g["spans"][g["lidxs"], 0]) > 1))
    bad[:, 0] = (bad[:, 0] + 1) % 4           # a column that exists nowhere in its locus
    for arr in (bad,):
        try:
            R.check_replicate(g["seqarr"], g["spans"], g["lidxs"], arr, g["tmpmap"])
        except AssertionError:
            continue
        raise AssertionError("checker accepted a corrupted replicate")


def test_driver_draw_order():
    """run_inference.py:117-123: choice(nloci), integers(2**31), integers(2**31) -- in that order."""
    g = load_golden("resample_T7_S300")
    rng = np.random.default_rng(5)
    _, _, lidxs, s1, s2 = R.resample_tmp_database(g["seqarr"], g["spans"], rng)
    chk = np.random.default_rng(5)
    nloci = g["spans"].shape[0]
    np.testing.assert_array_equal(lidxs, chk.choice(nloci, nloci, replace=True))
    assert s1 == int(chk.integers(2**31)) and s2 == int(chk.integers(2**31))
    assert rng.bit_generator.state == chk.bit_generator.state
