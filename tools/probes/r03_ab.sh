#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B of env-knob variants on one box.  Usage: r03_ab.sh "<flags>" VAR1=a VAR2=b ... (each arg one variant; "A=0" = baseline)
out=gpurun_out/r03_ab.txt
mkdir -p gpurun_out
flags=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-aggregate --no-events --steps 200 $flags > /tmp/b.json 2> /tmp/b.err
    echo "rep $rep $v -> $(python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])")" >> $out
  done
done
cat $out
