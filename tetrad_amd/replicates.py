"""Bootstrap replicates end to end on the GPUs: host-side mirror of the replicate loop of
tetrad/src/run_inference.py:378-407 (minus the supertree step, which is out of scope).

Per replicate the reference (1) resamples loci and rewrites the HDF5 database (:388-389,
`resample_tmp_database` :99-143), (2) draws a fresh random quartet sample from the project Generator
(`iter_chunks_random` :384 -- a generator, so its `rng.choice` runs at the first `next()` inside
`distributor`, i.e. AFTER the three draws of the resampler), (3) resolves the chunks on the engines and
collects the rows (:392).  Here:

  * the draws stay on ONE NumPy Generator in exactly that order -- locus indices, two integer seeds, the
    rank sample -- so the Generator state saved after replicate k (:405-407) is the reference's.  They do
    not depend on any result, so a producer thread makes them several replicates ahead
    (`ahead`), straight into page-locked buffers;
  * the replicate is built on the device right behind the previous replicate's kernels
    (`tq_bootstrap_async`: nothing but the locus indices crosses PCIe);
  * the sample is unranked on the device and resolved by `ShardedResolver` (every rank its parts, one
    all-gather per piece, rows to the destination rank's host arrays), two replicates in flight: the D2H of
    replicate k runs under the kernels of replicate k+1.

The host draw `rng.choice(C(T,4), Q, replace=False)` costs NumPy about 30 ms for 1e6 of 10.7e6 -- more than a
replicate's kernels on one GPU, and the same on every rank; `combinations.choice_without_replacement` makes the
same draws from the same bit generator about 3x faster (stream-identical, checked against NumPy at first use).
`sampler="device"` replaces it altogether by the library's
counter-based sampler (`tq_sample_quartets_dev`: same distribution, not the Generator's stream; the
Generator then only provides one extra integer seed per replicate).  Opt-in, like the device bootstrap's
own random streams it is documented as distribution-equal, not stream-identical.
"""
from __future__ import annotations

import queue
import threading
import time
from math import comb
from typing import Callable, Optional

import numpy as np

from . import bootstrap
from .combinations import choice_without_replacement
from .distributor import ShardedResolver
from .engine import QuartetEngine, pinned_empty


class ReplicateRunner:
    def __init__(self, engine: QuartetEngine, seqarr: np.ndarray, spans: np.ndarray, nquartets: int, *,
                 seed=None, rng: Optional[np.random.Generator] = None, sampler: str = "host", group=None,
                 pieces: Optional[int] = None, dst: Optional[int] = 0, ahead: int = 3, quartets_to_host: bool = False,
                 gather: str = "collective"):
        import torch
        if sampler not in ("host", "device"):
            raise ValueError("sampler must be 'host' or 'device'")
        self.torch = torch
        self.engine = engine
        self.sampler = sampler
        self.rng = rng if rng is not None else np.random.default_rng(seed)
        engine.set_source(seqarr, spans)
        self.nloci = int(np.asarray(spans).reshape(-1, 2).shape[0])
        self.T = int(seqarr.shape[0])
        self.total = comb(self.T, 4)
        self.Q = int(nquartets)
        if self.Q > self.total:
            raise ValueError(f"cannot sample {self.Q} of {self.total} quartets")
        # gather="host": the rows handed to on_result are views of a shared segment, valid until the replicate after next
        self.res = ShardedResolver(self.Q, engine=engine, group=group, pieces=pieces, dst=dst, nbuf=2, gather=gather)
        self.n_local = self.res.n_local
        self.ranges = self.res.ranges
        self.dev = self.res.dev
        self.ahead = max(1, int(ahead))
        self._q: Optional[queue.Queue] = None
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()
        self.rng_state_after: list = []          # Generator state after the draws of each consumed replicate
        n = max(1, self.n_local)
        self._d_ranks = torch.empty(n, dtype=torch.int64, device=self.dev)
        self._d_all = torch.empty((self.Q, 4), dtype=torch.int32, device=self.dev) if sampler == "device" else None
        self._local_index = torch.from_numpy(self.res.plan.local_index(self.res.rank)).to(self.dev) \
            if sampler == "device" else None
        self._d_q = self.res.quartet_buffer()
        # the whole sample of a replicate as quartets on the destination rank's host (for the supertree step):
        # unranked / sampled on the device into one of two buffers, copied out on the resolver's side stream
        self.want_quartets = bool(quartets_to_host) and (dst is None or dst == self.res.rank)
        self._d_full = [torch.empty((self.Q, 4), dtype=torch.int32, device=self.dev) for _ in range(2)] \
            if self.want_quartets else None
        self._d_ranks_full = torch.empty(self.Q, dtype=torch.int64, device=self.dev) \
            if (self.want_quartets and sampler == "host") else None

    # -- the draws of one replicate, in the reference's order ---------------------------------------------
    def _draw(self):
        rng = self.rng
        lidxs, s1, s2 = bootstrap.draw_replicate(self.nloci, rng)              # run_inference.py:117-123
        if self.sampler == "host":
            # combinations.py:113 -- NumPy's own sample and stream, through the library's faster restatement of its
            # tail shuffle when that applies (combinations.choice_without_replacement)
            idx = choice_without_replacement(rng, self.total, self.Q)
            local = pinned_empty(max(1, self.n_local), np.int64)
            o = 0
            for lo, hi in self.ranges:                                         # this rank's parts
                local[o:o + hi - lo] = idx[lo:hi]
                o += hi - lo
            sample = local
            if self.want_quartets:
                full = pinned_empty(self.Q, np.int64)
                full[...] = idx
                sample = (local, full)
        else:
            sample = int(rng.integers(2**63))                                   # one seed for the device sampler
        return lidxs, s1, s2, sample, rng.bit_generator.state

    def _producer(self, n: int):
        for _ in range(n):
            if self._stop.is_set():
                break
            t0 = time.perf_counter()
            d = self._draw()
            item = (d, (time.perf_counter() - t0) * 1e3)
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue

    # -- the loop -----------------------------------------------------------------------------------------
    def run(self, nreps: int, subsample_snps: bool = True, on_result: Optional[Callable] = None) -> dict:
        """`nreps` bootstrap replicates.  `on_result(k, S, rstat, rscor, flags)` -- with `quartets_to_host` also a
        sixth argument `quartets` u32[Q,4], the replicate's sample in row order -- is called on the destination rank as
        each replicate's rows arrive (host arrays owned by the callee).  Returns timing / flag stats."""
        torch, eng, res = self.torch, self.engine, self.res
        self._stop.clear()
        self._q = queue.Queue(maxsize=self.ahead)
        self._thread = threading.Thread(target=self._producer, args=(nreps,), daemon=True)
        self._thread.start()
        cur = torch.cuda.current_stream(self.dev)
        stats = dict(sites=[], host_ms=[], wait_ms=[], flags=dict(zero_data=0, degenerate=0, no_convergence=0))
        pending = None

        def collect(k, S, b):
            rstat, rscor, flags = res.finish(b)
            if flags is not None:
                stats["flags"]["zero_data"] += int((flags & 1).sum())
                stats["flags"]["degenerate"] += int(((flags & 2) != 0).sum())
                stats["flags"]["no_convergence"] += int(((flags & 8) != 0).sum())
                if on_result is not None:
                    if self.want_quartets:
                        on_result(k, S, rstat, rscor, flags, res.sets[b].pop("quartets"))
                    else:
                        on_result(k, S, rstat, rscor, flags)

        try:
            for k in range(nreps):
                t0 = time.perf_counter()
                (lidxs, s1, s2, sample, state), draw_ms = self._q.get()
                stats["wait_ms"].append((time.perf_counter() - t0) * 1e3)
                stats["host_ms"].append(draw_ms)
                self.rng_state_after.append(state)
                S = eng.bootstrap(lidxs, s1, s2, stream=cur.cuda_stream)        # enqueued behind replicate k-1
                stats["sites"].append(S)
                b = k & 1
                keep = None
                if self.sampler == "host":
                    local, full = sample if self.want_quartets else (sample, None)
                    keep = (local, full)                                        # alive until the copies have run
                    if self.n_local:
                        self._d_ranks.copy_(torch.from_numpy(local), non_blocking=True)
                        eng.unrank_dev(self._d_ranks.data_ptr(), self.n_local, self._d_q.data_ptr(), cur.cuda_stream)
                    if self.want_quartets:
                        self._d_ranks_full.copy_(torch.from_numpy(full), non_blocking=True)
                        eng.unrank_dev(self._d_ranks_full.data_ptr(), self.Q, self._d_full[b].data_ptr(), cur.cuda_stream)
                else:
                    d_all = self._d_full[b] if self.want_quartets else self._d_all
                    eng.sample_quartets_dev(sample, self.Q, d_all.data_ptr(), 0, cur.cuda_stream)
                    if self.n_local:
                        torch.index_select(d_all, 0, self._local_index, out=self._d_q)
                q_host = None
                if self.want_quartets:
                    q_host = pinned_empty((self.Q, 4), np.uint32)
                    if res.side is not None:
                        res.side.wait_stream(cur)
                        with torch.cuda.stream(res.side):
                            torch.from_numpy(q_host.view(np.int32)).copy_(self._d_full[b], non_blocking=True)
                    else:
                        q_host[...] = self._d_full[b].cpu().numpy().view(np.uint32)
                res.start(subsample_snps, b)
                res.sets[b]["keep"] = keep
                res.sets[b]["quartets"] = q_host
                if pending is not None:
                    collect(*pending)
                pending = (k, S, b)
            if pending is not None:
                collect(*pending)
        finally:
            self._stop.set()
            self._thread.join()
        return stats

    def close(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        self.res.close()


def bootstrap_trees(engine: QuartetEngine, seqarr: np.ndarray, spans: np.ndarray, nquartets: int, nboots: int, *,
                    subsample_snps: bool = True, weights: int = 0, min_snps: int = 0, min_ratio: float = 1.0, seed=None,
                    rng: Optional[np.random.Generator] = None, sampler: str = "host", group=None, workers: int = 4) -> list:
    """The bootstrap part of run_inference.py:378-407 including the supertree step (:394): `nboots` replicates through
    `ReplicateRunner`, each replicate's rows turned into a quartet supertree by the clean-room weighted Quartet MaxCut
    (`qmc.infer_supertree_from_arrays`: same filters and weight strategies as :254-305) on a small thread pool while
    the GPUs work on the next replicates.  Returns the newick strings in replicate order on the destination rank
    (rank 0), an empty list elsewhere.  Tip labels are taxon numbers (`qmc.relabel_tree` turns them into names)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import qmc
    runner = ReplicateRunner(engine, seqarr, spans, nquartets, seed=seed, rng=rng, sampler=sampler, group=group,
                             quartets_to_host=True)
    ntaxa = int(seqarr.shape[0])
    futures = {}
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        def on_result(k, S, rstat, rscor, flags, quartets):
            futures[k] = pool.submit(qmc.infer_supertree_from_arrays, quartets, rscor, rstat, ntaxa, weights, min_snps,
                                     min_ratio, k)
        try:
            runner.run(nboots, subsample_snps, on_result=on_result)
        finally:
            runner.close()
        return [futures[k].result() for k in sorted(futures)]
