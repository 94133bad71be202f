#!/bin/bash
# Run ON THE GPU BOX: where to start the target matching (MSL_MATCH_AFTER = block index; 0 = at the start of the step)
b() { timeout -k 10 200 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-aggregate $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for k in 0 1 2 3 4 5 0 3; do echo "== match after $k: $(MSL_MATCH_AFTER=$k b)"; done
for k in 0 2 3 4; do echo "== bf16 match after $k: $(MSL_MATCH_AFTER=$k b '--dtype bf16')"; done
