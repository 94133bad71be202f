"""gpurun_out/prof_<tag>/pmc_<set>.csv (rocprofv3 counter_collection, one row per dispatch and counter) -> pmc_avg.csv with one
row per kernel: launches and the average of every counter.  Runs on the GPU box (the raw files exceed the merge limit)."""
import csv
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(os.listdir(d)):
    if not (f.startswith("pmc_") and f.endswith(".csv")) or f == "pmc_avg.csv":
        continue
    for r in csv.DictReader(open(os.path.join(d, f))):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
    os.remove(os.path.join(d, f))
names = sorted({c for k in acc for c in acc[k]})
with open(os.path.join(d, "pmc_avg.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + names)
    for k in sorted(acc):
        n = max(cnt[k].values())
        w.writerow([k, n] + [f"{acc[k][c] / cnt[k][c]:.2f}" if cnt[k].get(c) else "" for c in names])
print("pmc_avg.csv:", len(acc), "kernels,", names)
