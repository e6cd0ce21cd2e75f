#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab7.log; : > $L; : > gpurun_out/r05_ab7.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-extras $args 2>>gpurun_out/r05_ab7.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('phase_ms_per_step') or {}; q=d['lu_panels']; print('   ms_per_step %.2f  big %.2f asm %.2f neither %.2f lane %.2f  res %.2e' % (d['ms_per_step'], p.get('big_updates',0), p.get('assembly_in_the_timed_region',0), p.get('stream_neither',0), p.get('lane_updates',0), d['check']['max_rel_residual']))" | tee -a $L
}
run "split 0, lanes high priority" "" MA_LU_CU_SPLIT=0 MA_LU_LANE_PRIO=1 &&
run "split 32, lanes high priority" "" MA_LU_CU_SPLIT=32 MA_LU_LANE_PRIO=1 &&
run "split 0, lanes high priority, 4 slots" "--slots 4" MA_LU_CU_SPLIT=0 MA_LU_LANE_PRIO=1 &&
run "split 32, 4 slots" "--slots 4" MA_LU_CU_SPLIT=32 &&
run "split 0, lanes high priority, kb 8" "" MA_LU_CU_SPLIT=0 MA_LU_LANE_PRIO=1 MA_LU_KB=8 &&
run "split 0, lanes high priority, block step" "" MA_LU_CU_SPLIT=0 MA_LU_LANE_PRIO=1 MA_LU_BLOCK_STEP=1
