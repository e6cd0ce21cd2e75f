#!/bin/bash
# under the round-3 default (pair panels + 64-CU split): spacing of the slots and panels per big update
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab14.log; : > $L; : > gpurun_out/r03_ab14.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab14.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "default" "" X=1 &&
run "spacing 12" "" MA_STAGE_SPACING=12 &&
run "spacing 13" "" MA_STAGE_SPACING=13 &&
run "spacing 14" "" MA_STAGE_SPACING=14 &&
run "spacing 16" "" MA_STAGE_SPACING=16 &&
run "kb 2" "" MA_LU_KB=2 &&
run "kb 3" "" MA_LU_KB=3 &&
run "kb 6" "" MA_LU_KB=6 &&
run "kb 8" "" MA_LU_KB=8 &&
run "default again" "" X=1
