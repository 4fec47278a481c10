"""Host logic of the multi-GPU path on CPU: sharding plan, world_size-2 gloo gather and regrouping,
TSV text.  The per-rank compute step is the oracle here (test infrastructure); on a GPU box the
default compute is the HIP engine (tests/test_gpu_parity.py covers that path)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import load_golden
from tetrad_amd import distributor as D

REPO = Path(__file__).resolve().parents[1]


def test_shard_bounds_cover_and_balance():
    for Q in (0, 1, 7, 8, 1820, 1_000_003):
        for w in (1, 2, 3, 8):
            b = D.shard_bounds(Q, w)
            assert b[0][0] == 0 and b[-1][1] == Q
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_get_chunksize_matches_reference_rule():
    # run_inference.py:73-96: chunk = nq // (breaks*ncores) + nq % (breaks*ncores)
    assert D.get_chunksize(1820, 4) == 1820 // 4 + 1820 % 4
    assert D.get_chunksize(635_376, 8) == 635_376 // (16 * 8) + 635_376 % (16 * 8)
    assert D.get_chunksize(6_000_000, 8) == 6_000_000 // (32 * 8) + 6_000_000 % (32 * 8)
    assert D.get_chunksize(3, 80) == 3


def test_shard_plan_covers_every_row_once():
    """Pieces x per-rank parts tile [0,Q) exactly; the gathered slab of a piece never reaches beyond the
    slack rows; the processing order of a rank is its parts in piece order."""
    for Q in (1, 2, 7, 101, 1820, 1_000_003):
        for w in (1, 2, 3, 8):
            for pieces in (None, 1, 2, 5, 8, 50):
                P = D.ShardPlan(Q, w, pieces)
                assert 1 <= P.npieces <= min(8, Q)
                seen = np.zeros(Q, np.int32)
                for r in range(w):
                    idx = P.local_index(r)
                    seen[idx] += 1
                    assert (np.diff(idx) > 0).all()
                assert (seen == 1).all()
                for i in range(P.npieces):
                    assert P.start[i] + w * P.part[i] <= P.rows_padded
                    assert w * P.part[i] >= P.end[i] - P.start[i]
                    assert P.slab_bytes(i) >= 33 * P.part[i] and P.slab_bytes(i) % 16 == 0
                    # part r of piece i holds global rows start + r*part + k
                    for r in range(w):
                        lo, hi = P.part_range(i, r)
                        assert lo == min(P.start[i] + r * P.part[i], P.end[i]) and lo <= hi <= P.end[i]


def test_piece_bounds_tile_the_batch_with_decreasing_sizes():
    """Result pieces are contiguous, cover [0,Q) once and shrink by PIECE_RATIO (the last piece's gather + D2H is what a
    step cannot hide); the automatic piece count keeps the last per-rank part above ~100k quartets."""
    for Q in (1, 5, 9, 1000, 1_000_000, 5_000_000, 174_792_640):
        for n in (1, 2, 3, 8):
            b = D.piece_bounds(Q, n)
            assert b[0][0] == 0 and b[-1][1] == Q and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            assert all(hi > lo for lo, hi in b) or Q < n
            sizes = [hi - lo for lo, hi in b]
            if Q >= 1000 * n and n > 1:
                assert all(sizes[i] > sizes[i + 1] for i in range(n - 1))
                assert abs(sizes[-1] / sizes[-2] - D.PIECE_RATIO) < 0.02
    for Q, world in ((1_000_000, 1), (8_000_000, 8), (5_000_000, 8), (5_000_000, 2), (40_000_000, 8)):
        P = D.ShardPlan(Q, world)
        assert P.npieces >= 1 and (P.npieces == 1 or P.part[-1] >= 100_000)
        assert P.npieces == 1 or P.part[0] >= 250_000


def test_single_process_resolver_with_injected_compute(oracle):
    """world = 1, host compute: the piece / regroup logic alone (several pieces, ragged last part)."""
    g = load_golden("tree_T12_S2000")
    q = g["quartets"][:97]

    def compute(ql, sub):
        _, rstat, rscor, dbg = oracle.new_infer_resolved_quartets(g["tmparr"], g["tmpmap"], ql, sub, debug=True)
        return rstat, rscor, dbg["flags"]

    want = compute(q, True)
    for pieces in (1, 3, 8):
        res = D.ShardedResolver(len(q), compute=compute, pieces=pieces)
        res.set_quartets(q)
        got = res.resolve(True)
        for a, b in zip(want, got):
            np.testing.assert_array_equal(a, b)


def test_tsv_text_equals_reference_pandas_call():
    """run_inference.py:233-234 builds the rows with pandas; the text must be identical."""
    import pandas as pd
    g = load_golden("c1_T16_S5000")
    rq, rstat, rscor = g["quartets"][:200], g["sub_rstat"][:200], g["sub_rscor"][:200]
    tabular = pd.concat([pd.DataFrame(i) for i in (rq, rscor, rstat)], axis=1)
    ref = tabular.to_csv(sep="\t", float_format='%.6f', index=False, header=False)
    assert D.format_tsv(rq, rscor, rstat) == ref


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import oracle as orc
from tetrad_amd import distributor as D
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "tree_T12_S2000.npz"))
def compute(tmparr, tmpmap, q, sub):
    _, rstat, rscor, dbg = orc.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
    return rstat, rscor, dbg["flags"]
qr = g["quartets"][:101]                       # odd count: exercises the padded slab
_, rstat, rscor, flags = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, True, compute=compute)
np.savez(out + f".{rank}.npz", rstat=rstat, rscor=rscor, flags=flags)
# no collective: every rank writes its rows into a shared host segment (gather="host"), rows to every rank / to rank 0
_, rs_h, rc_h, fl_h = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, True, compute=compute, pieces=2, gather="host")
np.savez(out + f".host.{rank}.npz", rstat=rs_h, rscor=rc_h, flags=fl_h)
_, rs_h0, rc_h0, fl_h0 = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, False, compute=compute, gather="host", dst=0)
if rank == 0:
    np.savez(out + ".host0.npz", rstat=rs_h0, rscor=rc_h0, flags=fl_h0)
else:
    assert rs_h0 is None and rc_h0 is None and fl_h0 is None
# several pieces (one all-gather each), rows to rank 0 only
_, rstat3, rscor3, flags3 = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, True, compute=compute, pieces=3, dst=0)
if rank == 0:
    np.savez(out + ".p3.npz", rstat=rstat3, rscor=rscor3, flags=flags3)
else:
    assert rstat3 is None and rscor3 is None and flags3 is None
# file-writing mirror of run_inference.distributor: only rank 0 writes
db = out + ".db.npz"
if rank == 0:
    np.savez(db, tmparr=g["tmparr"], tmpmap=g["tmpmap"])
dist.barrier()
chunks = [qr[i:i + 40].tolist() for i in range(0, 101, 40)]
D.distributor(db, out + ".tsv", 12, iter(chunks), True, None, compute=compute)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_two_rank_gloo_gather_equals_single_rank(tmp_path, oracle):
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port, out = _free_port(), str(tmp_path / "res")
    procs = [subprocess.Popen([sys.executable, str(script), str(REPO), str(r), "2", port, out])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    g = load_golden("tree_T12_S2000")
    _, rstat, rscor = oracle.new_infer_resolved_quartets(g["tmparr"], g["tmpmap"], g["quartets"][:101], True)
    for r in range(2):
        z = np.load(out + f".{r}.npz")
        np.testing.assert_array_equal(z["rstat"], rstat)              # N-rank == 1-rank, bitwise
        np.testing.assert_array_equal(z["rscor"], rscor)
    z = np.load(out + ".p3.npz")
    np.testing.assert_array_equal(z["rstat"], rstat)
    np.testing.assert_array_equal(z["rscor"], rscor)
    # gather="host": the shared-segment path gives the same rows, on every rank and on rank 0 alone (other mode)
    for r in range(2):
        z = np.load(out + f".host.{r}.npz")
        np.testing.assert_array_equal(z["rstat"], rstat)
        np.testing.assert_array_equal(z["rscor"], rscor)
    _, rstat_f, rscor_f = oracle.new_infer_resolved_quartets(g["tmparr"], g["tmpmap"], g["quartets"][:101], False)
    z = np.load(out + ".host0.npz")
    np.testing.assert_array_equal(z["rstat"], rstat_f)
    np.testing.assert_array_equal(z["rscor"], rscor_f)
    # rank 0 wrote all chunks in order, 9 columns
    rows = Path(out + ".tsv").read_text().splitlines()
    assert len(rows) == 101
    first = rows[0].split("\t")
    assert len(first) == 9 and [int(x) for x in first[:4]] == g["quartets"][0].tolist()
    assert rows[-1].split("\t")[7] == str(int(rstat[100, 0]))


SKEW_WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import oracle as orc
from tetrad_amd import distributor as D
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "tree_T12_S2000.npz"))
calls = []
def compute(tmparr, tmpmap, q, sub):
    t0 = time.perf_counter()
    _, rstat, rscor, dbg = orc.new_infer_resolved_quartets(tmparr, tmpmap, q, sub, debug=True)
    if rank == 1:                                  # this rank's compute step takes three times as long
        time.sleep(2.0 * (time.perf_counter() - t0) + 0.05)
    calls.append(len(q))
    return rstat, rscor, dbg["flags"]
qr = g["quartets"][:157]
res = {}
for step in range(3):                              # several steps back to back: a slow rank must not let pieces of
    for pieces in (1, 4):                          # different steps or different pieces overtake each other
        _, rstat, rscor, flags = D.resolve_sharded(g["tmparr"], g["tmpmap"], qr, bool(step & 1), compute=compute, pieces=pieces)
        res[f"rstat_{step}_{pieces}"] = rstat
        res[f"rscor_{step}_{pieces}"] = rscor
np.savez(out + f".{rank}.npz", calls=np.array(calls), **res)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gather_with_one_rank_three_times_slower(tmp_path, oracle):
    """Skewed per-rank cost ('optionally finer-grained tiles if per-quartet cost varies'): the partition is
    static, so a slow rank stretches the step -- but the piece pipeline must neither deadlock nor reorder: every rank
    ends with exactly the one-rank rows, for one piece and for four, over several consecutive steps in both modes."""
    import subprocess
    script = tmp_path / "skew_worker.py"
    script.write_text(SKEW_WORKER)
    port, out = _free_port(), str(tmp_path / "skew")
    procs = [subprocess.Popen([sys.executable, str(script), str(REPO), str(r), "2", port, out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    g = load_golden("tree_T12_S2000")
    want = {}
    for sub in (False, True):
        _, rstat, rscor = oracle.new_infer_resolved_quartets(g["tmparr"], g["tmpmap"], g["quartets"][:157], sub)
        want[sub] = (rstat, rscor)
    for r in range(2):
        z = np.load(out + f".{r}.npz")
        for step in range(3):
            for pieces in (1, 4):
                np.testing.assert_array_equal(z[f"rstat_{step}_{pieces}"], want[bool(step & 1)][0])
                np.testing.assert_array_equal(z[f"rscor_{step}_{pieces}"], want[bool(step & 1)][1])
        assert z["calls"].sum() > 0
