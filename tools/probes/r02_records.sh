#!/bin/bash
# A/B: how many event records the backward chain carries for the weight-gradient streams
mkdir -p gpurun_out/rec
for v in all "6,4,2" "5,3" "4" "none"; do
  if [ "$v" = all ]; then unset MSL_WGRAD_RECORD_AT; elif [ "$v" = none ]; then export MSL_WGRAD_RECORD_AT=""; else export MSL_WGRAD_RECORD_AT=$v; fi
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-aggregate --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['value'])" || exit 1
  done
done
