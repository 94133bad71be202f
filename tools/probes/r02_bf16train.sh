#!/bin/bash
# bf16 training path: kernel + model tests, then bench in both dtypes
set -o pipefail
mkdir -p gpurun_out/bf16t
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -s > gpurun_out/bf16t/tests.log 2>&1
rc=$?
tail -25 gpurun_out/bf16t/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/bf16t/bench_bf16.json 2> gpurun_out/bf16t/bench_bf16.err && \
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/bf16t/bench_f32.json 2> gpurun_out/bf16t/bench_f32.err
rc=$?
tail -3 gpurun_out/bf16t/bench_bf16.err; cut -c1-400 gpurun_out/bf16t/bench_bf16.json
tail -3 gpurun_out/bf16t/bench_f32.err; cut -c1-400 gpurun_out/bf16t/bench_f32.json
exit $rc
