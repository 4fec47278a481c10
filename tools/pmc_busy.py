#!/usr/bin/env python
"""Busy fractions of the hot kernels from `rocprofv3 --kernel-trace --pmc ...` passes (csv output), written
as profiles/pmc_busy_latest.json -- bench.py quotes them per kernel in roofline.kernels[].pmc.

    python tools/pmc_busy.py DIR [DIR ...] --out profiles/pmc_busy_latest.json --source profiles/r02_.../ --commit abc123

Definitions (per dispatch averages; SQ_ACTIVE_INST_* and SQ_BUSY_CU_CYCLES count quad-cycles summed over the chip):
  valu_busy      = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES        (share of CU-busy time with a VALU instruction executing,
                                                                   the 4 SIMDs of a CU counted together, as in r01's summaries)
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
            if "SQ_ACTIVE_INST_VALU" in m:
                e["valu_busy"] = round(m["SQ_ACTIVE_INST_VALU"] / busy, 4)
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
