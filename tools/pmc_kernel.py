"""Average the PMC counters of the kernels whose name contains a substring (rocprofv3 --pmc ... --output-format csv).
Usage: python tools/pmc_kernel.py <counter_collection.csv> <substring> [<substring> ...]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in rows:
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    print(f"== {pat}: " + ", ".join(f"{k}={acc[k] / cnt[k]:.4g}" for k in sorted(acc)) + f"  (launches {max(cnt.values()) if cnt else 0})")
