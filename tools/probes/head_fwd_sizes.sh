#!/bin/bash
# Run ON THE GPU BOX: the head kernels alone at the training shapes (16^3 / 8^3 / 4^3 x 4) and, forward only, at the 192^3
# inference shapes (24^3 / 12^3 x 2).
for cfg in "4 128 16 0" "4 256 8 0" "4 512 4 0" "2 256 24 1" "2 512 12 1"; do
  set -- $cfg
  HEAD_FWD_ONLY=$4 HEAD_N=$1 HEAD_C=$2 HEAD_D=$3 python tools/bench_head.py 2>&1 | grep "head"
done
