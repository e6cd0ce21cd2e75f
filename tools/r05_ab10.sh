#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab10.log; : > $L; : > gpurun_out/r05_ab10.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --warmup 6 --no-cpu-baseline --no-extras --no-check --no-timing $args 2>>gpurun_out/r05_ab10.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f  %s' % (d['ms_per_step'], d['config']['mode'][-110:]))" | tee -a $L
}
run "3 slots" "--steps 48" MA_LU_CU_SPLIT=32 &&
run "6 slots on 3 lanes" "--steps 48 --slots 6" MA_LU_CU_SPLIT=32 MA_LU_LANE_ALIAS=3 &&
run "5 slots on 3 lanes" "--steps 48 --slots 5" MA_LU_CU_SPLIT=32 MA_LU_LANE_ALIAS=3 &&
run "4 slots on 3 lanes" "--steps 48 --slots 4" MA_LU_CU_SPLIT=32 MA_LU_LANE_ALIAS=3 &&
run "4 slots on 2 lanes" "--steps 48 --slots 4" MA_LU_CU_SPLIT=32 MA_LU_LANE_ALIAS=2 &&
run "6 slots on 3 lanes, 20 steps" "--steps 20 --slots 6" MA_LU_CU_SPLIT=32 MA_LU_LANE_ALIAS=3 &&
run "6 slots on 3 lanes, split 0" "--steps 48 --slots 6" MA_LU_CU_SPLIT=0 MA_LU_LANE_ALIAS=3
