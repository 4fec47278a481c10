"""The N>1 product path with the real engine: two ranks (gloo collective, both on the one GPU of the
test box) resolve a batch through `distributor.resolve_sharded` / `distributor.distributor`; every rank
must end up with the rows of a single-engine run, bitwise, and rank 0 must write the same TSV.
(The "nccl" branch differs only in where the gathered tensor lives; it needs one GPU per rank.)"""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import load_golden

REPO = Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from tetrad_amd import distributor as D
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "c1_T16_S5000.npz"))
qr = g["quartets"][:1501]                      # odd count: exercises the padded slab
for sub in (True, False):
    _, rstat, rscor, flags = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, sub, pieces=(3 if sub else None))
    np.savez(out + f".{int(sub)}.{rank}.npz", rstat=rstat, rscor=rscor, flags=flags)
    # the alternative without a collective: every rank's rows D2H into a shared page-locked segment
    _, rstat, rscor, flags = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, sub, pieces=(2 if sub else None), gather="host")
    np.savez(out + f".host.{int(sub)}.{rank}.npz", rstat=rstat, rscor=rscor, flags=flags)
db = out + ".db.npz"
if rank == 0:
    np.savez(db, tmparr=g["tmparr"], tmpmap=g["tmpmap"])
dist.barrier()
chunks = [qr[i:i + 400].tolist() for i in range(0, 1501, 400)]
D.distributor(db, out + ".tsv", 16, iter(chunks), True, None)
dist.barrier()
# the replicate loop on two ranks: rows of every replicate to rank 0
from tetrad_amd import synth
from tetrad_amd.resolve_quartets import get_engine
from tetrad_amd.replicates import ReplicateRunner
seqarr, maparr, spans = synth.make_c5_source(T=14, S=6000, seed=8, ambiguous=0.02)
for sampler in ("host", "device"):
    got = {}
    runner = ReplicateRunner(get_engine(0), seqarr, spans, 701, seed=5, sampler=sampler, pieces=2,
                             gather=("host" if sampler == "device" else "collective"))
    runner.run(3, True, on_result=lambda k, S, a, b, c: got.__setitem__(k, (a.copy(), b.copy(), c.copy())))
    runner.close()
    if rank == 0:
        np.savez(out + f".reps.{sampler}.npz", **{f"{n}{k}": got[k][i] for k in got for i, n in enumerate(("rstat", "rscor", "flags"))})
    else:
        assert not got
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.gpu
def test_two_ranks_with_the_engine_equal_one_engine(tmp_path):
    from tetrad_amd import distributor as D
    from tetrad_amd.engine import QuartetEngine
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port, out = _free_port(), str(tmp_path / "res")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(REPO), str(r), "2", port, out], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    g = load_golden("c1_T16_S5000")
    q = g["quartets"][:1501]
    eng = QuartetEngine(0)
    eng.set_data(g["tmparr"], g["tmpmap"])
    for sub in (True, False):
        rstat, rscor, flags = eng.resolve(q, sub)
        for r in range(2):
            for tag in ("", ".host"):
                z = np.load(out + f"{tag}.{int(sub)}.{r}.npz")
                np.testing.assert_array_equal(z["rstat"], rstat)
                np.testing.assert_array_equal(z["rscor"], rscor)
                np.testing.assert_array_equal(z["flags"], flags)
    rstat, rscor, _ = eng.resolve(q, True)
    assert Path(out + ".tsv").read_text() == D.format_tsv(q, rscor, rstat)
    eng.close()
    # two-rank replicate loop == one-rank replicate loop, bitwise
    from tetrad_amd import synth
    from tetrad_amd.replicates import ReplicateRunner
    seqarr, maparr, spans = synth.make_c5_source(T=14, S=6000, seed=8, ambiguous=0.02)
    for sampler in ("host", "device"):
        one = {}
        with QuartetEngine(0) as e1:
            runner = ReplicateRunner(e1, seqarr, spans, 701, seed=5, sampler=sampler)
            runner.run(3, True, on_result=lambda k, S, a, b, c: one.__setitem__(k, (a.copy(), b.copy(), c.copy())))
            runner.close()
        z = np.load(out + f".reps.{sampler}.npz")
        for k in range(3):
            np.testing.assert_array_equal(z[f"rstat{k}"], one[k][0])
            np.testing.assert_array_equal(z[f"rscor{k}"], one[k][1])
            np.testing.assert_array_equal(z[f"flags{k}"], one[k][2])


@pytest.mark.gpu
def test_bare_bench_command_runs_two_ranks():
    """The driver's form of the command -- `python bench.py --gpus 2`, no launcher environment -- must start
    two ranks itself and report what the process group saw (gloo: both ranks share the one GPU of this box)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--quartets", "40000"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    # the same with --gather host (no collective on the data path)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--quartets", "40000", "--gather", "host", "--no-c4-leg", "--gather-ab"], capture_output=True,
                       text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert d["n_gpus"] == 2 and d["gather_verified"] and d["one_gpu_rows_equal_gathered_rows"]
    leg = d["other_gather_leg"]
    assert leg["gather"] == "collective" and leg["gather_verified"] and leg["rows_equal_main_leg"]
    assert d["config"]["quartets"] == 80000                      # weak: every rank its own 40 000
    assert d["gather_verified"] is True
    assert d["one_gpu_rows_equal_gathered_rows"] is True
    assert d["one_gpu_same_run_value"] > 0 and d["speedup"] > 0
    assert d["config"]["launcher"].startswith("bench.py started")
