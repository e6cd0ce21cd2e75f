#!/bin/bash
# early blocks give a share of their big update's columns to the slot's lane
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab21.log; : > $L; : > gpurun_out/r03_ab21.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab21.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
MA_LU_LANE_SHARE=15 MA_LU_LANE_SHARE_MIN_ROWS=300 timeout -k 10 300 python -m pytest tests/test_lu_gpu.py -q -x -k "staged" 2>&1 | tail -2 | tee -a $L
run "no share" "" X=1 &&
run "share 5 %, >= 6000 rows" "" MA_LU_LANE_SHARE=5 MA_LU_LANE_SHARE_MIN_ROWS=6000 &&
run "share 10 %, >= 6000 rows" "" MA_LU_LANE_SHARE=10 MA_LU_LANE_SHARE_MIN_ROWS=6000 &&
run "share 15 %, >= 6000 rows" "" MA_LU_LANE_SHARE=15 MA_LU_LANE_SHARE_MIN_ROWS=6000 &&
run "share 25 %, >= 6000 rows" "" MA_LU_LANE_SHARE=25 MA_LU_LANE_SHARE_MIN_ROWS=6000 &&
run "share 10 %, >= 4000 rows" "" MA_LU_LANE_SHARE=10 MA_LU_LANE_SHARE_MIN_ROWS=4000 &&
run "share 10 %, >= 8000 rows" "" MA_LU_LANE_SHARE=10 MA_LU_LANE_SHARE_MIN_ROWS=8000 &&
run "share 10 %, all blocks" "" MA_LU_LANE_SHARE=10 MA_LU_LANE_SHARE_MIN_ROWS=0
