#!/bin/bash
for v in 256 384 512 768 1024 2048; do
  echo "blocks/image $v: $(MSL_STEM_FWD_BLOCKS=$v timeout -k 10 120 python tools/bench_stem.py 2>/dev/null | head -1)"
done
