"""Does co-scheduling the scan of one half-batch with the singular-value stage of the other pay?  Two engines (each with its
own count slab and scratch) resolve half of the c3 batch each on their own stream, the second one started behind the first
one's scan; against one engine on the whole batch.  (Round 2 recorded 'two-stream overlap: no gain'; re-measured with round
4's kernels.)"""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from tetrad_amd import synth
from tetrad_amd.engine import QuartetEngine

T, S, _ = synth.CONFIGS["c3"]
tmparr, tmpmap = synth.simulate_tmparr(T, S, 103)
Q = 1_000_000
q = synth.random_quartets(T, Q, 4242)
dev = torch.device("cuda:0")
d_q = torch.from_numpy(q.view(np.int32)).to(dev)
out = lambda n: (torch.zeros((n, 2), dtype=torch.int32, device=dev), torch.zeros((n, 3), dtype=torch.float64, device=dev),
                 torch.zeros(n, dtype=torch.uint8, device=dev))
e1, e2 = QuartetEngine(0), QuartetEngine(0)
e1.set_data(tmparr, tmpmap); e2.set_data(tmparr, tmpmap)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
oa = out(Q)
def one():
    e1.resolve_dev(d_q.data_ptr(), Q, True, oa[0].data_ptr(), oa[1].data_ptr(), oa[2].data_ptr(), s1.cuda_stream)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"one engine, 1e6 quartets: {timeit(one):.3f} ms")
for nparts in (2, 4):
    h = Q // nparts
    outs = [out(h) for _ in range(nparts)]
    def split():
        for k in range(nparts):
            e, s = (e1, s1) if k % 2 == 0 else (e2, s2)
            o = outs[k]
            e.resolve_dev(d_q.data_ptr() + k * h * 16, h, True, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), s.cuda_stream)
    print(f"two engines on two streams, {nparts} parts of {h}: {timeit(split):.3f} ms")
    same = all(torch.equal(outs[k][0], oa[0][k * h:(k + 1) * h]) and torch.equal(outs[k][1], oa[1][k * h:(k + 1) * h]) for k in range(nparts))
    print("   rows equal:", same)
