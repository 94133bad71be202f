#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects everything the profiles/ summaries are made from into
# gpurun_out/prof_<tag>/ .  Usage: tools/collect_profiles.sh <tag>
#   1. bench.py (default flags, with the CPU baseline)            -> bench.json / bench.err
#   2. rocprofv3 --kernel-trace --stats of the same workload       -> kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two passes, counters only + kernel trace) -> pmc_*.csv
set -e
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python3 "$root/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-aggregate > "$out/ks.log" 2>&1
cp "$(find /tmp/ks -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- python3 "$root/bench.py" --steps 6 --warmup 6 --no-cpu-baseline --no-aggregate > "$out/pmc_$c.log" 2>&1
  cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" "$out/pmc_$c.csv"
done
ls -la "$out"
