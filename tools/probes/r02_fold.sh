#!/bin/bash
# A/B: BatchNorm fold threshold of the depthwise consumers (per-layer depthwise durations and step time)
for v in 65536 512 64 32; do
  export MSL_FOLD_NP_MAX=$v
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); a=d['roofline']['depthwise_fwd_all_layers']
print('$v', d['ms_per_step'], d['value'], 'dw1 event us', d['roofline']['avg_launch_us'], 'in-step', a['in_step_event_pairs']['sum_launch_us'], a['in_step_event_pairs']['per_layer_us'], 'b2b', a['back_to_back']['sum_launch_us'], a['back_to_back']['per_layer_us'])" || exit 1
done
