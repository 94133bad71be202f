#!/bin/bash
# Run ON THE GPU BOX from the repo root: configs[3] (192^3 x 2 inference, bench.py --mode infer) -> gpurun_out/prof_<tag>/
#   bench.json / bench.err, kernel_stats.csv + kernel_trace.csv (rocprofv3 --kernel-trace --stats of the same command),
#   pmc_avg.csv (FETCH_SIZE / WRITE_SIZE / MFMA-busy passes).  Usage: tools/collect_infer_profile.sh <tag> [--dtype bf16]
set -e
tag=${1:-r03_infer}
shift || true
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
python bench.py --mode infer "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "bench done: $(cut -c1-200 "$out/bench.json")"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o ks --output-format csv -- python3 "$root/bench.py" --mode infer --steps 40 --warmup 10 --no-cpu-baseline --map-cases 2 "$@" > "$out/ks.log" 2>&1
cp "$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
python "$root/tools/timeline.py" "$(find /tmp/ks_$tag -name '*kernel_trace.csv' | head -1)" -2 stem_dw_eval > "$out/timeline.txt" 2>&1 || true
n=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_${tag}_$n
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_${tag}_$n -o p --output-format csv -- python3 "$root/bench.py" --mode infer --steps 6 --warmup 6 --no-cpu-baseline --map-cases 2 "$@" > "$out/pmc_$name.log" 2>&1
  cp "$(find /tmp/pmc_${tag}_$n -name '*counter_collection.csv' | head -1)" "$out/pmc_$name.csv"
  n=$((n+1))
done
cd "$root"
python tools/pmc_average.py "$out"
ls "$out"
