#!/usr/bin/env python
"""Host<->device copy rates on this box: torch-pinned vs the library's pinned pool vs pageable arrays."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tetrad_amd.engine import pinned_empty

dev = torch.device("cuda:0")
for mb in (4, 16, 32, 128):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    srcs = {"torch-pinned": torch.empty(n, dtype=torch.uint8).pin_memory(),
            "pool-pinned": torch.from_numpy(pinned_empty(n, np.uint8)),
            "pageable": torch.empty(n, dtype=torch.uint8)}
    for name, h in srcs.items():
        h.fill_(1)
        for direction in ("h2d", "d2h"):
            def go():
                if direction == "h2d":
                    d.copy_(h, non_blocking=True)
                else:
                    h.copy_(d, non_blocking=True)
            go(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                go()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(f"{mb:4d} MiB {name:13s} {direction}: {dt*1e3:7.3f} ms  {n/dt/1e9:6.1f} GB/s  (is_pinned={h.is_pinned()})")
