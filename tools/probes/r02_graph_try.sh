for g in 0 1; do
export MSL_USE_GRAPH=$g
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-aggregate --no-events > gpurun_out/r02_graph$g.json 2> gpurun_out/r02_graph$g.err
echo graph=$g rc=$?
tail -3 gpurun_out/r02_graph$g.err
python -c "
import json; d=json.loads(open('gpurun_out/r02_graph$g.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
