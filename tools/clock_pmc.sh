#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, from one rocprofv3 pass
# (MI355X_MICROARCH.md, "DVFS give-back").   tools/clock_pmc.sh OUTDIR [bench.py arguments]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O; rm -rf $O/clk
timeout -k 5 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/clk --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 3 --warmup 1 "$@" > /dev/null 2> $O/clk.err
python3 - $O/clk <<'PY'
import csv, sys
from pathlib import Path
from collections import defaultdict
d = Path(sys.argv[1])
acc = defaultdict(lambda: [0.0, 0.0, 0])
for f in d.rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f, newline="")):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if dur < 3e5: continue                      # the quotient reads high on short dispatches
        n = r["Kernel_Name"]; k = n[n.find("tq_"):][:24] if "tq_" in n else n[:24]
        a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
for k, (c, t, n) in sorted(acc.items()):
    print(f"{k:26s} dispatches {n:4d}  mean {t / n / 1e6:7.3f} ms  effective clock {c / 8 / t:5.2f} GHz")
PY
rm -rf $O/clk
