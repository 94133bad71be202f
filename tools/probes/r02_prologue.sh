#!/bin/bash
# Run ON THE GPU BOX: forward prologue (NaN-flag reset, head weight packing, optimiser hyper-parameter copy) on the heads stream
b() { timeout -k 10 200 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-aggregate $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in 1 0 1 0 1; do echo "== f32 prologue_on_side=$v: $(MSL_PROLOGUE_ON_SIDE=$v b)"; done
for v in 1 0; do echo "== bf16 prologue_on_side=$v: $(MSL_PROLOGUE_ON_SIDE=$v b '--dtype bf16')"; done
for v in 1 0; do echo "== 2ch prologue_on_side=$v: $(MSL_PROLOGUE_ON_SIDE=$v b '--channels 2')"; done
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py tests/test_gpu_parallel.py -x -q -m gpu 2>&1 | tail -3
for dt in f32 bf16; do echo "== infer $dt: $(timeout -k 10 200 python tools/bench_infer.py --dtype $dt 2>&1 | grep predict_step)"; done
