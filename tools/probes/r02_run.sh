python -m pytest tests -m gpu -x -q > gpurun_out/r02_t6.log 2>&1; tail -4 gpurun_out/r02_t6.log
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-aggregate"
for t in 1 2; do
MSL_ENQUEUE_THREADS=$t $B > gpurun_out/r02_mt$t.json 2> gpurun_out/r02_mt$t.err
echo threads=$t $(tail -1 gpurun_out/r02_mt$t.err) $(python -c "
import json; d=json.loads(open('gpurun_out/r02_mt$t.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
done
MSL_FOLD_NP_MAX=4096 $B > gpurun_out/r02_fold.json 2> gpurun_out/r02_fold.err
echo fold4096 $(tail -1 gpurun_out/r02_fold.err) $(python -c "
import json; d=json.loads(open('gpurun_out/r02_fold.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
MSL_FOLD_NP_MAX=4096 MSL_FOLD_NP_MAX_PW=64 $B > gpurun_out/r02_fold2.json 2> gpurun_out/r02_fold2.err
echo fold4096+pw64 $(tail -1 gpurun_out/r02_fold2.err) $(python -c "
import json; d=json.loads(open('gpurun_out/r02_fold2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
