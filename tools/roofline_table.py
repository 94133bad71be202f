"""profiles/<tag>_kernel_stats.csv + <tag>_pmc_summary.csv -> profiles/<tag>_roofline_table.md: per kernel average duration,
HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction) and achieved bandwidth against the 8 TB/s peak.
Usage: python tools/roofline_table.py <tag>"""
import csv
import os
import re
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv")))}
pmc = list(csv.DictReader(open(os.path.join(root, "profiles", f"{tag}_pmc_summary.csv"))))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


rows = []
for r in pmc:
    st = stats.get(r["kernel"])
    if st is None:
        continue
    us = float(st["AverageNs"]) / 1e3
    mb = (2.0 * float(r["FETCH_SIZE_kb_avg_raw"]) + float(r["WRITE_SIZE_kb_avg"])) * 1024 / 1e6
    if "at::native" in r["kernel"] or "rocclr" in r["kernel"]:
        continue
    rows.append((float(st["TotalDurationNs"]), short(r["kernel"]), int(st["Calls"]), us, mb, mb / us if us else 0.0))  # MB/us == TB/s
rows.sort(reverse=True)
steps = max(int(stats[k]["Calls"]) for k in stats if "adam_kernel" in k)
out = [f"# Per-kernel HBM traffic and bandwidth ({tag}; 128^3 x 4 training step, all three streams running)", "",
       "Average rocprofv3 duration, PMC traffic per launch (`FETCH_SIZE` x2 + `WRITE_SIZE`, separate passes), achieved = traffic / "
       "duration, fraction of the 8 TB/s HBM3E peak.  Kernels below ~20 us are latency-, not bandwidth-bound.", "",
       "| kernel | launches/step | avg us | HBM MB/launch | TB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
for _, n, calls, us, mb, tbs in rows[:24]:
    out.append(f"| `{n}` | {calls / steps:.1f} | {us:.1f} | {mb:.1f} | {tbs:.2f} | {tbs / 8.0:.2f} |")
open(os.path.join(root, "profiles", f"{tag}_roofline_table.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
