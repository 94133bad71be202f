#!/bin/bash
root=$(pwd); cd /tmp && export TMPDIR=/tmp
for mode in alone busy; do
  rm -rf /tmp/dv_$mode
  rocprofv3 --kernel-trace --stats -d /tmp/dv_$mode -o dv --output-format csv -- python3 "$root/tools/probes/r02_scan_dvfs.py" $mode > /tmp/dv_$mode.log 2>&1
  f=$(find /tmp/dv_$mode -name '*kernel_stats.csv' | head -1)
  echo "== $mode"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "detect_" in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']:>4}  {r['Name'].split('(')[1].split('::')[-1] if '::' in r['Name'] else r['Name'][:40]}")
PY
done
