"""Scan of one part under the singular-value stage of the previous one, STAGGERED: two engines on two streams, the batch cut
into parts; stream 2 starts its first scan when stream 1's first scan has ended, so that from then on a scan (LDS / integer
issue) and a singular-value stage (f64 issue) are in flight together.  Against one engine on the whole batch."""
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
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
pads = [int(a) for a in sys.argv[1:]] or [0]
for sub, pad in [(True, p) for p in pads] + [(False, 0)]:
    if pad:            # (needs a build with the experiment's `scan_pad_lds` option: unused dynamic LDS on the scan's launch, i.e.
        e1.set_option("scan_pad_lds", pad); e2.set_option("scan_pad_lds", pad)      # fewer scan workgroups per CU; not in the product)
    print(f"-- scan_pad_lds {pad}")
    def one():
        e1.resolve_dev(d_q.data_ptr(), Q, sub, oa[0].data_ptr(), oa[1].data_ptr(), oa[2].data_ptr(), s1.cuda_stream)
    print(f"sub={sub} one engine, 1e6 quartets: {timeit(one):.3f} ms", flush=True)
    for nparts in (2, 4, 8):
        h = Q // nparts
        outs = [out(h) for _ in range(nparts)]
        def split():
            ev = torch.cuda.Event()
            for k in range(nparts):
                e, s = (e1, s1) if k % 2 == 0 else (e2, s2)
                o = outs[k]
                if k == 1:
                    s2.wait_event(ev)
                e.scan_dev(d_q.data_ptr() + k * h * 16, h, sub, s.cuda_stream)
                if k == 0:
                    ev.record(s1)
                e.svd_dev(0, h, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), s.cuda_stream)
        print(f"sub={sub} staggered, {nparts} parts of {h}: {timeit(split):.3f} ms", flush=True)
        same = all(torch.equal(outs[k][0], oa[0][k * h:(k + 1) * h]) and torch.equal(outs[k][1], oa[1][k * h:(k + 1) * h]) for k in range(nparts))
        print("   rows equal:", same)
