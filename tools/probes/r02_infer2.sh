#!/bin/bash
# Run ON THE GPU BOX: detection / predict_step parity after the host-path changes, then the inference timings
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py -x -q -m gpu -k "detect or 192 or infer or predict or eval" 2>&1 | tail -3
for dt in f32 bf16; do
  echo "== infer $dt resident: $(timeout -k 10 200 python tools/bench_infer.py --dtype $dt 2>&1 | grep predict_step)"
  echo "== infer $dt staged  : $(timeout -k 10 200 python tools/bench_infer.py --dtype $dt --staged-copy 2>&1 | grep predict_step)"
done
