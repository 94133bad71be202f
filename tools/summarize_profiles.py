"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
  profiles/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.csv and dw_fwd1_traffic.json.
Usage: python tools/summarize_profiles.py <tag>"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))


def per_kernel(path, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


fetch = per_kernel(os.path.join(src, "pmc_FETCH_SIZE.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(src, "pmc_WRITE_SIZE.csv"), "WRITE_SIZE")
rows = sorted(((k, fetch[k][1], fetch[k][0], write.get(k, (0.0, 0))[0]) for k in fetch), key=lambda r: -(r[2] * 2 + r[3]))
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_kb_avg_raw,WRITE_SIZE_kb_avg,note: FETCH_SIZE x2 for 16-B/lane streaming reads on gfx950\n")
    for k, n, fk, wk in rows[:40]:
        f.write(f'"{k}",{n},{fk:.1f},{wk:.1f},\n')
# block 1's depthwise forward: the stride-2 wave kernel on 64^2 planes (the streamed kernel before r01_d)
key = [k for k in fetch if "dw_s2_wave_kernel<4, 5, 4>" in k] or [k for k in fetch if "dw_fwd_stream_kernel<2, 1, 4, 0>" in k]
if key:
    fk, wk = fetch[key[0]][0], write[key[0]][0]
    hbm = (2.0 * fk + wk) * 1024.0
    json.dump({
        "kernel": re.search(r"(\w+<[^>]*>)", key[0]).group(1).replace(" ", ""), "round": 1, "profile": tag, "FETCH_SIZE_kb_avg": fk, "WRITE_SIZE_kb_avg": wk,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads -> x2 "
                      "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 150998400,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                  "--steps 6 --warmup 6 --no-cpu-baseline (tools/collect_profiles.sh)",
        "note": "above the algorithmic bytes by the halo input plane shared by adjacent 4-plane slabs (12.5 % of the "
                "planes), as far as the XCD L2 / Infinity Cache does not serve it",
    }, open(os.path.join(dst, "dw_fwd1_traffic.json"), "w"), indent=1)
    print(f"dw_fwd1 traffic: {hbm / 1e6:.1f} MB per launch ({hbm / 150998400:.3f} x algorithmic)")
print(open(os.path.join(dst, f"{tag}_bench.json")).read()[:600])
