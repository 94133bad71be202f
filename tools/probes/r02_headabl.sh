#!/bin/bash
for v in 0 1 2 3 4; do echo "ablation $v: $(MSL_HEAD_ABL=$v timeout -k 10 120 python tools/bench_head.py 2>/dev/null | head -1)"; done
python tools/bench_head.py 2>/dev/null | tail -2
