#!/bin/bash
# Run ON THE GPU BOX: event-record sharing / host-thread A-B on the chain-bound step (each line: knobs -> ms/step, host enqueue)
out=gpurun_out/r03_records.txt
mkdir -p gpurun_out
: > $out
run() {
  env "$@" python bench.py --no-cpu-baseline --no-aggregate --no-events --steps 200 > /tmp/b.json 2> /tmp/b.err
  echo "$* -> $(python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])") | $(tail -n 1 /tmp/b.err)" >> $out
}
run A=0
run MSL_WGRAD_RECORD_AT=6,4,2,1
run MSL_WGRAD_RECORD_AT=5,1
run MSL_WGRAD_RECORD_AT=1
run MSL_ENQUEUE_THREADS=1
run MSL_MATCH_AFTER=5
run MSL_WGRAD_LAG=2
run A=1
cat $out
