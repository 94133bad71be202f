#!/bin/bash
# Run ON THE GPU BOX: parity of the eval-mode rows depthwise kernel + 192^3 inference A/B
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -x -q -m gpu -k "eval_rows or dw_fwd or dw_bwd" 2>&1 | tail -3
for dt in f32 bf16; do
  echo "== infer $dt rows: $(timeout -k 10 200 python tools/bench_infer.py --dtype $dt 2>&1 | grep predict_step)"
  echo "== infer $dt lds : $(MSL_DW_ROWS_EVAL=0 timeout -k 10 200 python tools/bench_infer.py --dtype $dt 2>&1 | grep predict_step)"
done
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py -x -q -m gpu -k "192 or infer or predict or eval" 2>&1 | tail -3
bash tools/prof_infer.sh r02_infer_f32e > /dev/null 2>&1; head -12 gpurun_out/r02_infer_f32e/stats_short.txt
bash tools/prof_infer.sh r02_infer_bf16e --dtype bf16 > /dev/null 2>&1; head -12 gpurun_out/r02_infer_bf16e/stats_short.txt
