#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B of schedule options on ONE box (boxes differ by up to 5 % in step time, so only
# same-box comparisons count).  Usage: r03_ab.sh "<bench flags>" "<variant 1>" "<variant 2>" ...   where a variant is a list
# of bench.py flags, e.g. "--opt match_after=5" or "" for the default; three interleaved repeats; results in gpurun_out/r03_ab.txt
out=gpurun_out/r03_ab.txt
mkdir -p gpurun_out
flags=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    python bench.py --no-cpu-baseline --no-aggregate --no-events --steps 200 $flags $v > /tmp/b.json 2> /tmp/b.err
    echo "rep $rep [$v] -> $(python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])")" >> $out
  done
done
cat $out
