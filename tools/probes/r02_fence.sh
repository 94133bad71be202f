#!/bin/bash
# Run ON THE GPU BOX: per-step round trip through the caller's stream (MSL_BENCH_FENCE=1) vs steps enqueued back to back
b() { timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aggregate $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in 0 1 0 1; do echo "== f32 fence=$v: $(MSL_BENCH_FENCE=$v b)"; done
for v in 0 1; do echo "== f32 fence=$v prologue on chain: $(MSL_PROLOGUE_ON_SIDE=0 MSL_BENCH_FENCE=$v b)"; done
for v in 0 1; do echo "== bf16 fence=$v: $(MSL_BENCH_FENCE=$v b '--dtype bf16')"; done
for v in 0 1; do echo "== 2ch fence=$v: $(MSL_BENCH_FENCE=$v b '--channels 2')"; done
echo "== default bench line:"; timeout -k 10 300 python bench.py 2>/dev/null | cut -c1-330
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parallel.py -x -q -m gpu 2>&1 | tail -2
