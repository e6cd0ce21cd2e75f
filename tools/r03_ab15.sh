#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab15.log; : > $L; : > gpurun_out/r03_ab15.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab15.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "kb 5" "" MA_LU_KB=5 &&
run "kb 5 spacing 10" "" MA_LU_KB=5 MA_STAGE_SPACING=10 &&
run "kb 6 spacing 8" "" MA_LU_KB=6 MA_STAGE_SPACING=8 &&
run "kb 6 spacing 9" "" MA_LU_KB=6 MA_STAGE_SPACING=9 &&
run "kb 6 spacing 10" "" MA_LU_KB=6 MA_STAGE_SPACING=10 &&
run "kb 7" "" MA_LU_KB=7 &&
run "kb 7 spacing 8" "" MA_LU_KB=7 MA_STAGE_SPACING=8 &&
run "kb 8 spacing 7" "" MA_LU_KB=8 MA_STAGE_SPACING=7 &&
run "kb 6" "" MA_LU_KB=6
