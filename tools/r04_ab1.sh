#!/bin/bash
# round 4, item 1(a): CU split x kb x spacing re-swept at the final round-3 kernels (one box, 48 steps each)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab1.log; : > $L; : > gpurun_out/r04_ab1.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 200 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r04_ab1.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "default (split 64, kb 6, G/3)" "" X=1 &&
run "split 24" "" MA_LU_CU_SPLIT=24 &&
run "split 32" "" MA_LU_CU_SPLIT=32 &&
run "split 40" "" MA_LU_CU_SPLIT=40 &&
run "split 48" "" MA_LU_CU_SPLIT=48 &&
run "split 56" "" MA_LU_CU_SPLIT=56 &&
run "split 48 kb 5" "" MA_LU_CU_SPLIT=48 MA_LU_KB=5 &&
run "split 48 kb 7" "" MA_LU_CU_SPLIT=48 MA_LU_KB=7 &&
run "split 40 kb 5" "" MA_LU_CU_SPLIT=40 MA_LU_KB=5 &&
run "split 40 kb 7" "" MA_LU_CU_SPLIT=40 MA_LU_KB=7 &&
run "split 64 kb 5" "" MA_LU_KB=5 &&
run "split 64 kb 7" "" MA_LU_KB=7 &&
run "split 64 kb 8" "" MA_LU_KB=8 &&
run "split 56 kb 7" "" MA_LU_CU_SPLIT=56 MA_LU_KB=7 &&
run "split 64 spacing 8" "" MA_STAGE_SPACING=8 &&
run "split 64 spacing 10" "" MA_STAGE_SPACING=10 &&
run "split 48 spacing 8" "" MA_LU_CU_SPLIT=48 MA_STAGE_SPACING=8 &&
run "default again" "" X=1
