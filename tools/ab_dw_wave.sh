#!/bin/bash
# A/B of depthwise-forward settings inside the full training step (alternating runs on the same box).
# Usage: tools/ab_dw_wave.sh "ENV=.. ENV=.." "ENV=.." ...   (each argument = one configuration; "-" = defaults)
set -e
run() { env $1 python bench.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('$1', d['value'], d['ms_per_step'], d['roofline'].get('depthwise_fwd_all_layers',{}).get('back_to_back',{}).get('per_layer_us'))"; }
for r in 1 2 3; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then run "MSL_NOOP=1"; else run "$cfg"; fi
  done
done
