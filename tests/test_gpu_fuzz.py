"""A short seeded run of tests/fuzz_gpu.py (random shapes, sparsity, locus structure, batch sizes
and engine options; GPU result vs oracle) inside the GPU suite.  Longer runs: `python tests/fuzz_gpu.py 600 <seed>`."""
import subprocess
import sys
from pathlib import Path

import pytest


@pytest.mark.gpu
def test_seeded_fuzz_run():
    script = Path(__file__).with_name("fuzz_gpu.py")
    r = subprocess.run([sys.executable, str(script), "20", "12345"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "cases clean" in r.stdout
