#!/bin/bash
for v in 256 512 768 1024 1536; do echo "blocks $v: $(MSL_STEM_BWW_BLOCKS=$v timeout -k 10 120 python tools/bench_stem.py 2>/dev/null | sed -n 2,3p | tr '\n' ' ')"; done
