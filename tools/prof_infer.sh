#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the inference workload (tools/bench_infer.py) -> gpurun_out/<tag>/
set -e
tag=${1:-prof_infer}; shift || true
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ki_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ki_$tag -o ki --output-format csv -- python3 "$root/tools/bench_infer.py" --steps 30 "$@" > "$out/ki.log" 2>&1
cp "$(find /tmp/ki_$tag -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
cd "$root"
python - "$out/kernel_stats.csv" <<'PY' > "$out/stats_short.txt"
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", n); return (m.group(1) if m else n)[:70]
for r in rows[:25]:
    print(f"{int(r['TotalDurationNs'])/1e3:10.1f} us total  x{int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f}  {short(r['Name'])}")
PY
head -16 "$out/stats_short.txt"
