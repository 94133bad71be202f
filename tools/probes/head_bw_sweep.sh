#!/bin/bash
# Run ON THE GPU BOX: forward head convolution, register-fed kernel (bw=0) against the LDS-staged kernel with 4- / 8- / 16-wide
# blocks, per map size.  The block width was forced by a TEMPORARY patch at the top of head_fwd_lds_bw (csrc/heads.hip),
#     if (const char* f = getenv("MSL_TMP_HEAD_BW")) { const int bw = atoi(f); ... return bw if it tiles the map, else 0; }
# which is not in the library (no environment switches there): re-apply it to repeat the sweep.  Results of round 3 are in the
# comment above head_fwd_lds_bw.
for cfg in "2 256 24" "2 512 12" "4 128 16" "4 256 8" "2 128 48" "1 128 32" "2 64 32"; do
  set -- $cfg
  for bw in 0 4 8 16; do
    echo -n "N=$1 C=$2 D=$3 bw=$bw: "
    HEAD_FWD_ONLY=1 HEAD_N=$1 HEAD_C=$2 HEAD_D=$3 MSL_TMP_HEAD_BW=$bw python tools/bench_head.py 2>&1 | tail -n 1
  done
done
