#!/bin/bash
# Run ON THE GPU BOX: parity of the tile-staged stem weight gradient, then A/B against the wave-private kernel.
out=gpurun_out/s2; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -x -q -m gpu -k "stem" > $out/t.log 2>&1; tail -3 $out/t.log
for c in 1 2; do
  echo "== bench_stem cin=$c tile"; timeout -k 10 100 python tools/bench_stem.py $c 2>&1 | tail -3
  echo "== bench_stem cin=$c wave"; MSL_STEM_BWW_TILE=0 timeout -k 10 100 python tools/bench_stem.py $c 2>&1 | tail -2
done
for a in "" "--dtype bf16" "--channels 2"; do
  echo "== bench $a tile"; timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-aggregate $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== bench $a wave"; MSL_STEM_BWW_TILE=0 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-aggregate $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
