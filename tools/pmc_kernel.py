"""Average PMC counter values per dispatch of the kernels whose name contains a pattern, from
`rocprofv3 --kernel-trace --pmc ... --output-format csv` directories (one directory per pass)."""
import csv, sys
from collections import defaultdict
from pathlib import Path

pat = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
for d in sys.argv[2:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f, newline="")):
            if pat in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k][0] / acc[k][1]:16.1f}  (n={acc[k][1]})")
