"""Per-kernel totals of a rocprofv3 --kernel-trace --stats run: python3 profiles/kstats.py <output directory> [filter]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if flt in name:
        v.sort()
        print(f"{name[:70]:70s} calls {len(v):5d}  total {sum(v):10.1f} us  median {v[len(v) // 2]:8.2f}  min {v[0]:8.2f}  max {v[-1]:8.2f}")
