"""bench.py --gpus N without a launcher environment must start its N ranks itself (the driver's bare
`python bench.py --gpus N`), relay rank 0's one JSON line and pass failures on.  CPU test of the launcher
and the rendezvous through `--launch-probe` (no GPU work); the real N-rank benchmark runs in
tests/test_gpu_distributed.py on the GPU box."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def run_bench(*argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *argv], capture_output=True, text=True, env=env,
                          timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_bare_command_starts_n_ranks(n):
    r = run_bench("--gpus", str(n), "--launch-probe", "--backend", "gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["gpus_arg"] == n        # ranks the process group saw
    assert d["rank_sum"] == n * (n + 1) // 2              # every rank took part in the collective
    assert d["self_launched"] and d["master_addr"] == "127.0.0.1"


def test_one_gpu_is_one_process():
    r = run_bench("--gpus", "1", "--launch-probe")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and not d["self_launched"]


def test_a_dead_rank_ends_the_job_with_a_failure():
    r = run_bench("--gpus", "2", "--launch-probe", "--backend", "gloo", env_extra={"TQ_BENCH_PROBE_FAIL_RANK": "1"},
                  timeout=120)
    assert r.returncode != 0
    assert "rank 1 exited with status 7" in r.stderr


def test_nccl_with_too_few_gpus_is_refused_not_downgraded():
    # this container has no GPU: asking for 2 RCCL ranks must fail loudly, never run a smaller job
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = run_bench("--gpus", "2", "--no-cpu")
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr
    assert r.stdout.strip() == ""


def test_mismatched_launcher_environment_is_refused():
    r = run_bench("--gpus", "4", "--launch-probe", env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    # an external launcher with another world size: main() must not silently run something else
    assert r.returncode != 0 or json.loads(r.stdout.strip())["n_gpus"] != 4


def test_a_terminated_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to the launching process (a driver's timeout) must end the ranks it started: none may stay behind."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TQ_BENCH_PROBE_SLEEP"] = "60"
    pidfile = tmp_path / "pids"
    env["TQ_BENCH_PROBE_PIDFILE"] = str(pidfile)
    p = subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--launch-probe", "--backend", "gloo"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.time() + 60
    while time.time() < deadline and (not pidfile.exists() or len(pidfile.read_text().split()) < 2):
        time.sleep(0.2)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) != 0
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive, f"rank process {pid} survived its launcher"
