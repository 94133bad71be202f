#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects everything the profiles/ summaries are made from into
# gpurun_out/prof_<tag>/ .  Usage: tools/collect_profiles.sh <tag> [extra bench.py flags, e.g. --channels 2 | --dtype bf16]
#   1. bench.py (with the CPU baseline unless the extra flags say otherwise)      -> bench.json / bench.err
#   2. rocprofv3 --kernel-trace --stats of the same workload                       -> kernel_stats.csv, kernel_trace.csv
#   3. rocprofv3 --pmc passes, counters + kernel trace only (one pass per set):    -> pmc_<set>.csv
#        FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
set -e
tag=${1:-rXX}
shift || true
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
python bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "bench done: $(cut -c1-160 "$out/bench.json")"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o ks --output-format csv -- python3 "$root/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-aggregate "$@" > "$out/ks.log" 2>&1
cp "$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
cp "$(find /tmp/ks_$tag -name '*kernel_trace.csv' | head -1)" "$out/kernel_trace.csv"
echo "kernel trace done"
n=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_${tag}_$n
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_${tag}_$n -o p --output-format csv -- python3 "$root/bench.py" --steps 6 --warmup 6 --no-cpu-baseline --no-aggregate "$@" > "$out/pmc_$name.log" 2>&1
  cp "$(find /tmp/pmc_${tag}_$n -name '*counter_collection.csv' | head -1)" "$out/pmc_$name.csv"
  echo "pmc pass $name done"
  n=$((n+1))
done
cd "$root"
python tools/timeline.py "$out/kernel_trace.csv" > "$out/timeline.txt" 2>&1 || true
# the raw per-dispatch counter files are large: keep per-kernel averages only
python tools/pmc_average.py "$out"
rm -f "$out"/pmc_*.csv.raw
ls -la "$out"
