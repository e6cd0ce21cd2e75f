#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab17.log; : > $L; : > gpurun_out/r03_ab17.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab17.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "default (DMA update kernel, kb 6, spacing 9)" "" X=1 &&
run "spacing 8" "" MA_STAGE_SPACING=8 &&
run "spacing 10" "" MA_STAGE_SPACING=10 &&
run "kb 8" "" MA_LU_KB=8 &&
run "kb 8 spacing 6" "" MA_LU_KB=8 MA_STAGE_SPACING=6 &&
run "kb 8 spacing 7" "" MA_LU_KB=8 MA_STAGE_SPACING=7 &&
run "kb 5" "" MA_LU_KB=5 &&
run "split 56" "" MA_LU_CU_SPLIT=56 &&
run "split 48" "" MA_LU_CU_SPLIT=48 &&
run "4 slots" "--slots 4" MA_BENCH_MAX_SLOTS=8
