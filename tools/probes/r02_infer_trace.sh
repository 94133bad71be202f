#!/bin/bash
# per-launch durations of one inference batch (192^3 x 2): tools/probes/r02_infer_trace.sh [--dtype bf16]
root=$(pwd); cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kit
rocprofv3 --kernel-trace -d /tmp/kit -o k --output-format csv -- python3 "$root/tools/bench_infer.py" --steps 6 "$@" > /tmp/kit.log 2>&1
python3 - "$(find /tmp/kit -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "detect_finalize" in r["Kernel_Name"]]
a, b = ends[-2], ends[-1]
t0 = int(rows[a + 1]["Start_Timestamp"])
for r in rows[a + 1:b + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0][:60]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {n}")
print(f"span {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us")
PY
