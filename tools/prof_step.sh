#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace + stats of the default bench workload -> gpurun_out/<tag>/{kernel_stats.csv,kernel_trace.csv,timeline.txt}
set -e
tag=${1:-prof}
shift || true
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o ks --output-format csv -- python3 "$root/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-aggregate "$@" > "$out/ks.log" 2>&1
cp "$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
cp "$(find /tmp/ks_$tag -name '*kernel_trace.csv' | head -1)" "$out/kernel_trace.csv"
cd "$root"
python tools/timeline.py "$out/kernel_trace.csv" > "$out/timeline.txt" 2>&1 || true
python - "$out/kernel_stats.csv" <<'PY' > "$out/stats_short.txt"
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", n); return (m.group(1) if m else n)[:70]
steps = max(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])
tot = 0
for r in rows:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    per = int(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    print(f"{per:8.1f} us/step  x{int(r['Calls'])/steps:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f}  {short(r['Name'])}")
print(f"total kernel time per step: {tot:.1f} us over {steps} steps")
PY
rm -f "$out/kernel_trace.csv.tmp"
tail -3 "$out/stats_short.txt"
