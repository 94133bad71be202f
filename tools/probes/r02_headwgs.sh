#!/bin/bash
# A/B: workgroups per launch of the LDS-staged head kernels (occupancy vs. partial-slab traffic)
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-aggregate --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['value'])" || exit 1; }
export MSL_HEAD_FWD_WGS=256 MSL_HEAD_BWD_WGS=256 MSL_HEAD_BWW_WGS=256; run "fwd256 bwd256 bww256"; run "fwd256 bwd256 bww256"
export MSL_HEAD_FWD_WGS=512; run "fwd512 bwd256 bww256"
export MSL_HEAD_FWD_WGS=768; run "fwd768(=1024) bwd256 bww256"
export MSL_HEAD_FWD_WGS=256 MSL_HEAD_BWD_WGS=512; run "fwd256 bwd512 bww256"
export MSL_HEAD_BWD_WGS=768; run "fwd256 bwd768 bww256"
export MSL_HEAD_BWD_WGS=256 MSL_HEAD_BWW_WGS=512; run "fwd256 bwd256 bww512"
export MSL_HEAD_FWD_WGS=512 MSL_HEAD_BWD_WGS=512 MSL_HEAD_BWW_WGS=512; run "all512"
export MSL_HEAD_FWD_WGS=512 MSL_HEAD_BWD_WGS=768 MSL_HEAD_BWW_WGS=256; run "fwd512 bwd768 bww256"
