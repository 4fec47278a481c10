"""Synthetic SNP matrices and quartet samples for tests and bench.py.

Produces inputs in exactly the layout the reference worker reads from its
HDF5 database:

  tmparr  u8 [T,S]  0=A 1=C 2=G 3=T 78=N   (tetrad/src/write_database.py:157-168)
  tmpmap  u32[S,2]  col0 = locus ordinal (non-decreasing), col1 = site index
                    (tetrad/src/write_database.py:138-149)
  quartets u32[Q,4] strictly increasing taxon indices, lexicographic in full mode
                    (tetrad/src/combinations.py:40-55, 94-121)

Generator: random coalescent-like binary tree on T tips,
root base uniform on {0..3}, per-branch substitution probability ``p`` with a
JC-style redraw, only sites variable among all T taxa kept, ``missing``
fraction of cells set to 78, loci = contiguous runs of length 1+Poisson(4).
"""
from __future__ import annotations

from itertools import combinations
from math import comb

import numpy as np

CONFIG_SEEDS = {"c1": 101, "c2": 102, "c3": 103, "c4": 104, "c5": 105}
CONFIGS = {
    # name: (T, S, Q or None for full)
    "c1": (16, 5_000, None),
    "c2": (64, 20_000, None),
    "c3": (128, 50_000, 1_000_000),
    "c4": (256, 100_000, 5_000_000),
}


def random_tree_children(T: int, rng: np.random.Generator):
    """Random-joining (coalescent-shaped) binary tree.  Returns (children, root):
    children[node] = (left, right) for internal nodes T..2T-2; tips are 0..T-1."""
    active = list(rng.permutation(T))
    children = {}
    nxt = T
    while len(active) > 1:
        i, j = rng.choice(len(active), size=2, replace=False)
        a, b = active[i], active[j]
        for k in sorted((i, j), reverse=True):
            active.pop(k)
        children[nxt] = (a, b)
        active.append(nxt)
        nxt += 1
    return children, active[0]


def simulate_tmparr(T: int, S: int, seed: int, p: float = 0.05, missing: float = 0.10):
    """Return (tmparr u8[T,S], tmpmap u32[S,2])."""
    rng = np.random.default_rng(seed)
    children, root = random_tree_children(T, rng)
    out = np.empty((T, 0), dtype=np.uint8)
    while out.shape[1] < S:
        n = max(1024, int((S - out.shape[1]) * 1.3) + 64)
        states = {root: rng.integers(0, 4, size=n, dtype=np.uint8)}
        stack = [root]
        tips = np.empty((T, n), dtype=np.uint8)
        while stack:
            node = stack.pop()
            st = states.pop(node)
            if node < T:
                tips[node] = st
                continue
            for ch in children[node]:
                cs = st.copy()
                mut = rng.random(n) < p
                cs[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
                states[ch] = cs
                stack.append(ch)
        variable = (tips != tips[0]).any(axis=0)
        out = np.concatenate([out, tips[:, variable]], axis=1)
    tmparr = np.ascontiguousarray(out[:, :S])
    tmparr[rng.random(tmparr.shape) < missing] = 78
    # loci: contiguous runs, lengths 1 + Poisson(4)
    lens = 1 + rng.poisson(4, size=S)
    locus = np.repeat(np.arange(S, dtype=np.uint32), lens)[:S]
    tmpmap = np.empty((S, 2), dtype=np.uint32)
    tmpmap[:, 0] = locus
    tmpmap[:, 1] = np.arange(S, dtype=np.uint32)
    return tmparr, tmpmap


def make_c5_source(T: int | None = None, S: int | None = None, seed: int | None = None, ambiguous: float = 0.01):
    """The project-level inputs of the bootstrap flow (BASELINE.json configs[4], c3 shape): `seqarr`
    u8[T,S] of ASCII bases with `ambiguous` of the cells replaced by IUPAC two-base codes and N for
    missing (write_database.py:157-159), the snpsmap-style `maparr` u32[S,2] and the locus spans
    i64[nloci,2] (jit/get_spans.py)."""
    T0, S0, _ = CONFIGS["c3"]
    T, S = T or T0, S or S0
    tmparr, tmpmap = simulate_tmparr(T, S, CONFIG_SEEDS["c5"] if seed is None else seed)
    ascii_ = np.array([65, 67, 71, 84], np.uint8)
    seqarr = np.where(tmparr <= 3, ascii_[np.minimum(tmparr, 3)], 78).astype(np.uint8)
    rs = np.random.default_rng(0)
    amb = rs.random(seqarr.shape) < ambiguous
    seqarr[amb] = rs.choice(np.array([82, 75, 83, 89, 87, 77], np.uint8), size=int(amb.sum()))
    loc = tmpmap[:, 0]
    starts = np.flatnonzero(np.concatenate([[True], loc[1:] != loc[:-1]]))
    spans = np.stack([starts, np.concatenate([starts[1:], [S]])], axis=1).astype(np.int64)
    return seqarr, tmpmap, spans


def all_quartets(T: int) -> np.ndarray:
    """All C(T,4) quartets in lexicographic order (combinations.py:40-55)."""
    n = comb(T, 4)
    out = np.fromiter((x for q in combinations(range(T), 4) for x in q),
                      dtype=np.uint32, count=4 * n)
    return out.reshape(n, 4)


def unrank_quartets(index: np.ndarray, T: int) -> np.ndarray:
    """Vectorised lexicographic unranking, same mapping as
    combinations.py:94-106 (_index_to_combination) for k=4."""
    index = np.asarray(index, dtype=np.int64).copy()
    Q = index.shape[0]
    out = np.zeros((Q, 4), dtype=np.uint32)
    nsel = np.zeros(Q, dtype=np.int64)
    for i in range(T):
        need = 4 - nsel - 1                     # remaining picks after this one
        active = nsel < 4
        # comb(T-i-1, need) for need in 0..3
        cvals = np.array([comb(T - i - 1, k) for k in range(4)], dtype=np.int64)
        c = cvals[np.clip(need, 0, 3)]
        take = active & (c > index)
        skip = active & ~take
        rows = np.nonzero(take)[0]
        out[rows, nsel[rows]] = i
        nsel[rows] += 1
        index[skip] -= c[skip]
    return out


def random_quartets(T: int, Q: int, seed: int) -> np.ndarray:
    """combinations.py:109-114: rng.choice(C(T,4), size=Q, replace=False) then
    lexicographic unranking.  Order is the sampled order (not sorted)."""
    rng = np.random.default_rng(seed)
    idx = rng.choice(comb(T, 4), size=Q, replace=False)
    return unrank_quartets(idx, T)


def make_config(name: str, Q: int | None = None):
    """Return (tmparr, tmpmap, quartets) for a BASELINE.json config."""
    T, S, q_default = CONFIGS[name]
    seed = CONFIG_SEEDS[name]
    tmparr, tmpmap = simulate_tmparr(T, S, seed)
    nq = q_default if Q is None else Q
    if nq is None:
        quartets = all_quartets(T)
    else:
        quartets = random_quartets(T, nq, seed + 1000)
    return tmparr, tmpmap, quartets
