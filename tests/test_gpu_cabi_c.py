"""GPU: a plain C host (examples/c_api_demo.c, gcc) drives the C ABI with no Python or torch in the
process; its TSV rows must equal what the oracle gives for the same generated input."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


def lcg_stream(seed):
    s = seed
    while True:
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        yield s >> 8


def make_input():
    T, S = 9, 3000
    g = lcg_stream(12345)
    tmparr = np.zeros((T, S), np.uint8)
    tmpmap = np.zeros((S, 2), np.uint32)
    locus = 0
    for s in range(S):
        anc = next(g) & 3
        for t in range(T):
            r = next(g) % 100
            tmparr[t, s] = 78 if r < 8 else ((next(g) & 3) if r < 40 else anc)
        if s and next(g) % 4 == 0:
            locus += 1
        tmpmap[s] = (locus, s)
    return tmparr, tmpmap


@pytest.mark.parametrize("sub", [1, 0])
def test_c_host_matches_oracle(tmp_path, oracle, sub):
    from tetrad_amd import _lib, distributor, synth
    exe = tmp_path / "c_api_demo"
    subprocess.check_call(["gcc", "-O2", str(REPO / "examples" / "c_api_demo.c"), f"-I{REPO / 'include'}",
                           f"-L{_lib.CSRC}", "-ltetrad_hip", f"-Wl,-rpath,{_lib.CSRC}", "-o", str(exe)])
    out = subprocess.run([str(exe), str(sub)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "rc=-1" in out.stderr                       # the bad-index call was refused with TQ_ERR_INVALID_ARG
    tree = [ln for ln in out.stderr.splitlines() if ln.startswith("tree: ")]
    assert len(tree) == 1 and tree[0].split()[1].endswith(";")      # wQMC lines + supertree from plain C
    assert sorted(int(x) for x in __import__("re").findall(r"\d+", tree[0].split()[1])) == list(range(9))
    tmparr, tmpmap = make_input()
    q = synth.all_quartets(9)
    _, rstat, rscor, dbg = oracle.new_infer_resolved_quartets(tmparr, tmpmap, q, bool(sub), debug=True)
    rows = out.stdout.splitlines()
    assert len(rows) == 126
    got = np.array([[float(x) for x in r.split("\t")] for r in rows])
    np.testing.assert_array_equal(got[:, :4], q)
    np.testing.assert_array_equal(got[:, 8], rstat[:, 1])
    ok = (dbg["flags"] & 3) == 0
    np.testing.assert_array_equal(got[ok, 7], rstat[ok, 0])
    assert np.abs(got[:, 4:7] - rscor).max() <= 1e-6 * max(1.0, np.abs(rscor).max())
    # and the text itself is what the reference's pandas call would write for the oracle's rows
    ref_rows = distributor.format_tsv(q, rscor, rstat).splitlines()
    same = sum(a == b for a, b in zip(rows, ref_rows))
    assert same >= 120      # identical rows except where the 6th decimal of a score rounds differently
