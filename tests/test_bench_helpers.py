"""CPU tests of bench.py's bookkeeping for the round-4 scan kernels: the unit count of the joint-histogram scan (what the
device's unit list must hold: tests/test_gpu_configs.py checks the kernel, this the model) and which kernel the per-kernel
roofline entries name."""
import importlib.util
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", REPO / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_module"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_unit_count_pairs_neighbours_inside_runs_of_equal_abc():
    bench = _bench()
    q = np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5],          # a run of three: one pair + one on its own
                  [0, 1, 3, 4],                                      # alone
                  [1, 2, 3, 4], [1, 2, 3, 5], [1, 2, 3, 6], [1, 2, 3, 7],   # a run of four: two pairs
                  [4, 3, 2, 1], [4, 3, 2, 0]], np.uint32)            # taxa in any order: the key is positional
    rng = np.random.default_rng(0)
    u = bench.dp_unit_count(q[rng.permutation(len(q))], T=8)
    assert u == {"pairs": 4, "singles": 2, "units": 6}
    # repeated quartets are one long run
    assert bench.dp_unit_count(np.repeat(q[:1], 7, axis=0), T=8) == {"pairs": 3, "singles": 1, "units": 4}


def test_roofline_entries_name_the_kernel_that_ran():
    bench = _bench()
    kms = {"order": 0.2, "scan": 5.4, "bidiag": 1.3, "bdsqr": 2.2, "score": 0.1}
    sub = bench.kernel_rooflines(kms, 1, 1_000_000, 50_000, True, "c3", None, True)
    assert [k["name"] for k in sub][1] == "tq_scan_f4_kernel"
    dp = {"pairs": 424_626, "singles": 150_748, "units": 575_374, "quartets": 1_000_000}
    full = bench.kernel_rooflines(kms, 1, 1_000_000, 50_000, False, "c3", dp, False)
    scan = full[1]
    assert scan["name"] == "tq_scan_dp_kernel" and abs(scan["units"]["quartets_per_wave_step"] - 1.738) < 1e-3
    old = bench.kernel_rooflines(kms, 1, 1_000_000, 50_000, False, "c3", None, False)
    assert old[1]["name"] == "tq_scan_wg_kernel"
    for entries in (sub, full, old):
        assert all(0 < k["lds_path_frac"] < 1.5 for k in entries if "lds_path_frac" in k)
