#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab6.log; : > $L; : > gpurun_out/r05_ab6.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-extras $args 2>>gpurun_out/r05_ab6.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('phase_ms_per_step') or {}; q=d['lu_panels']; print('   ms_per_step %.2f  big %.2f asm %.2f neither %.2f lane %.2f  res %.2e  acc %.1f wid %.1f rej %.1f' % (d['ms_per_step'], p.get('big_updates',0), p.get('assembly_in_the_timed_region',0), p.get('stream_neither',0), p.get('lane_updates',0), d['check']['max_rel_residual'], q['half_panels_accepted_per_step'], q['half_panels_accepted_widened_per_step'], q['half_panels_rejected_per_step']))" | tee -a $L
}
run "split 32" "" MA_LU_CU_SPLIT=32 &&
run "split 32 block step" "" MA_LU_CU_SPLIT=32 MA_LU_BLOCK_STEP=1 &&
run "split 24 block step" "" MA_LU_CU_SPLIT=24 MA_LU_BLOCK_STEP=1 &&
run "split 16 block step" "" MA_LU_CU_SPLIT=16 MA_LU_BLOCK_STEP=1 &&
run "split 32 block step kb 8" "" MA_LU_CU_SPLIT=32 MA_LU_BLOCK_STEP=1 MA_LU_KB=8 &&
run "split 32 block step kb 4" "" MA_LU_CU_SPLIT=32 MA_LU_BLOCK_STEP=1 MA_LU_KB=4 &&
run "split 32 spacing 8" "" MA_LU_CU_SPLIT=32 MA_STAGE_SPACING=8 &&
run "split 32 spacing 10" "" MA_LU_CU_SPLIT=32 MA_STAGE_SPACING=10 &&
run "split 40" "" MA_LU_CU_SPLIT=40 &&
run "split 40 block step" "" MA_LU_CU_SPLIT=40 MA_LU_BLOCK_STEP=1
