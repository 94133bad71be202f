#!/bin/bash
for v in 0 1 2 3 4 5; do echo "ablation $v: $(MSL_STEMB_ABL=$v timeout -k 10 120 python tools/bench_stem.py 2>/dev/null | sed -n 2p)"; done
