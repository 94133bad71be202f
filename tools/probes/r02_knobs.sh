set -e
mkdir -p gpurun_out/r02_knobs
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-aggregate"
$B > gpurun_out/r02_knobs/base.json 2> gpurun_out/r02_knobs/base.err
for v in 64 128 256 512 4096; do MSL_FOLD_NP_MAX=$v MSL_FOLD_NP_MAX_PW=32 $B > gpurun_out/r02_knobs/fold_$v.json 2> gpurun_out/r02_knobs/fold_$v.err; done
for v in 64 128; do MSL_FOLD_NP_MAX=32 MSL_FOLD_NP_MAX_PW=$v $B > gpurun_out/r02_knobs/foldpw_$v.json 2> gpurun_out/r02_knobs/foldpw_$v.err; done
$B > gpurun_out/r02_knobs/base2.json 2> gpurun_out/r02_knobs/base2.err
grep -h ms_per_step gpurun_out/r02_knobs/*.json | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])
"
for f in gpurun_out/r02_knobs/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
"; done
