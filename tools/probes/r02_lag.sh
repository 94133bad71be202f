python -m pytest tests -m gpu -x -q > gpurun_out/r02_t4.log 2>&1; tail -4 gpurun_out/r02_t4.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-aggregate"
for lag in 1 2 3 8; do
MSL_WGRAD_LAG=$lag $B > gpurun_out/r02_lag$lag.json 2> gpurun_out/r02_lag$lag.err
echo lag=$lag $(tail -1 gpurun_out/r02_lag$lag.err) $(python -c "
import json; d=json.loads(open('gpurun_out/r02_lag$lag.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
done
MSL_FOLD_NP_MAX=4096 $B > gpurun_out/r02_fold.json 2> gpurun_out/r02_fold.err
echo fold4096 $(tail -1 gpurun_out/r02_fold.err) $(python -c "
import json; d=json.loads(open('gpurun_out/r02_fold.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
bash tools/prof_step.sh r02_p2 > /dev/null 2>&1; sort -rn gpurun_out/r02_p2/stats_short.txt | head -12
