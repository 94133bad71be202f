#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "batch" 2>&1 | tail -2
b() { timeout -k 10 200 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-aggregate $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "== both on: $(b)"
echo "== pw batch off: $(MSL_PW_BWW_BATCH=0 b)"
echo "== gpack batch off: $(MSL_HEAD_GPACK_BATCH=0 b)"
echo "== both off: $(MSL_PW_BWW_BATCH=0 MSL_HEAD_GPACK_BATCH=0 b)"
echo "== both on again: $(b)"
echo "== bf16 on: $(b '--dtype bf16')"
echo "== bf16 gpack off: $(MSL_HEAD_GPACK_BATCH=0 b '--dtype bf16')"
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py tests/test_gpu_parallel.py -x -q -m gpu 2>&1 | tail -2
