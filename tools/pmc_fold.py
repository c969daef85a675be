"""Fold rocprofv3 counter_collection CSVs (any counters) into per-kernel averages per launch.  usage: pmc_fold.py <dir> [name filter]"""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set); dur = defaultdict(float)
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            n = row["Kernel_Name"]
            if filt and filt not in n: continue
            n = n[:70]
            acc[n][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (path, row["Dispatch_Id"])
            if key not in disp[n]:
                disp[n].add(key); dur[n] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for n in acc:
    k = len(disp[n])
    print(f"{n}  launches {k}  avg {dur[n] / k / 1e3:.1f} us")
    for c, v in sorted(acc[n].items()):
        print(f"    {c:36s} {v / k:16.1f} per launch")
