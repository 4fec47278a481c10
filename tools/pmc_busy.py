#!/usr/bin/env python
"""Busy fractions of the hot kernels from `rocprofv3 --kernel-trace --pmc ...` passes (csv output), written
as profiles/pmc_busy_latest.json -- bench.py quotes them per kernel in roofline.kernels[].pmc.

    python tools/pmc_busy.py DIR [DIR ...] --out profiles/pmc_busy_latest.json --source profiles/r02_.../ --commit abc123

Definitions (per dispatch averages, summed over the chip):
  valu_inst_per_cu_cycle = SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES      vector wave-instructions per CU-busy cycle.  A wave64 instruction on
                                                                   32-bit operands issues over 2 cycles of its SIMD (32 lanes per cycle,
                                                                   MI355X_MICROARCH.md), an f64 one over 4: a CU's 4 SIMDs take at most 2
                                                                   (1 for f64) per cycle
  valu_busy      = valu_inst_per_cu_cycle / that peak             (rounds 1-3 divided by 1 for every kernel, which made the integer scan
                                                                   kernels look twice as VALU-busy as they are: 0.76 was 0.38)
  lds_busy       = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import argparse
import csv
import json
from collections import defaultdict
from pathlib import Path

KERNELS = {"tq_scan_wg_kernel": "tq_scan_wg", "tq_scan_f4_kernel": "tq_scan_f4", "tq_scan_dp_kernel": "tq_scan_dp", "tq_bidiag_kernel": "tq_bidiag",
           "tq_bdsqr_kernel": "tq_bdsqr", "tq_score_kernel": "tq_score"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+", type=Path)
    ap.add_argument("--out", type=Path, required=True)
    ap.add_argument("--source", default="")
    ap.add_argument("--commit", default="unknown")
    args = ap.parse_args()
    acc = {k: defaultdict(lambda: [0.0, 0]) for k in KERNELS}
    for d in args.dirs:
        for f in d.rglob("*counter_collection.csv"):
            for row in csv.DictReader(open(f, newline="")):
                for k, pat in KERNELS.items():
                    if pat in row["Kernel_Name"]:
                        a = acc[k][row["Counter_Name"]]
                        a[0] += float(row["Counter_Value"])
                        a[1] += 1
    out = {"source": args.source, "commit": args.commit, "kernels": {}}
    for k, c in acc.items():
        m = {n: v[0] / v[1] for n, v in c.items() if v[1]}
        if not m:
            continue
        e = {"counters_per_dispatch": {n: round(v, 1) for n, v in sorted(m.items())}}
        busy = m.get("SQ_BUSY_CU_CYCLES")
        if busy:
            nv = m.get("SQ_INSTS_VALU", m.get("SQ_ACTIVE_INST_VALU"))
            if nv is not None:
                peak = 2.0 if k.startswith("tq_scan") else 1.0          # 32-bit integer work / f64 work
                e["valu_inst_per_cu_cycle"] = round(nv / busy, 4)
                e["valu_peak_inst_per_cu_cycle"] = peak
                e["valu_busy"] = round(nv / busy / peak, 4)
            if "SQ_LDS_IDX_ACTIVE" in m:
                e["lds_busy"] = round(m["SQ_LDS_IDX_ACTIVE"] / busy, 4)
            if "TA_TA_BUSY_sum" in m:       # texture-address path: every vector memory instruction passes it (tools/probe_ta.hip)
                e["ta_busy"] = round(m["TA_TA_BUSY_sum"] / busy, 4)
        if m.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in m:
            e["lds_conflict_share"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        out["kernels"][k] = e
    args.out.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: {x: v[x] for x in v if x != "counters_per_dispatch"} for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
