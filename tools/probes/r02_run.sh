python -m pytest tests -m gpu -x -q > gpurun_out/r02_t9.log 2>&1; tail -4 gpurun_out/r02_t9.log
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-aggregate"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r02_m_$tag.json 2> gpurun_out/r02_m_$tag.err; echo $tag $(tail -1 gpurun_out/r02_m_$tag.err | cut -c1-60) $(python -c "
import json; d=json.loads(open('gpurun_out/r02_m_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"); }
run base A=1
run base2 A=1
python tools/probes/alone.py 128 2>&1 | grep "head\|sum alone\|match"
