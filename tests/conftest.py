import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


GOLDEN_FULL_CASES = [
    "dense_T8_S400", "tree_T12_S2000", "sparse_T10_S257", "edge_T7_S130",
    "carry_T6_S2500", "tiny_T5_S37", "one_site_T5_S1", "lowrank_T9_S700", "minrank_T14_S600",
]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
