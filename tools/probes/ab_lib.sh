for rep in 1 2 3; do
  for v in old new; do
    cp tools/probes/_ab/lib$v.so mslesions3d_amd/libmsl3d_hip.so
    echo -n "rep $rep [$v] train: "; python bench.py --no-cpu-baseline --no-aggregate --no-events --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
    echo -n "rep $rep [$v] infer: "; python bench.py --mode infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
cp tools/probes/_ab/libnew.so mslesions3d_amd/libmsl3d_hip.so
