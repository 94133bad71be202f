#!/bin/bash
# A/B/C of the depthwise forward variants inside the full training step (alternating, same box):
#   A: MSL_DW_WAVE=1 (stride-1 wave kernel only)   B: default (stride 1 + 2)   C: B + block 1 on the stride-2 wave kernel
set -e
run() { env "$@" python bench.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'], d['roofline'].get('depthwise_fwd_all_layers',{}).get('per_layer_us'))"; }
for r in 1 2 3; do
  run MSL_DW_WAVE=1
  run MSL_DW_WAVE=3
  run MSL_DW_WAVE=3 MSL_DW_WAVE_S2_MAXW=64
done
