#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab18.log; : > $L; : > gpurun_out/r03_ab18.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab18.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
MA_LU_TAIL_ROWS=3000 timeout -k 10 300 python -m pytest tests/test_lu_gpu.py -q -x -k "staged or pipeline" 2>&1 | tail -2 | tee -a $L
run "no tail" "" MA_LU_TAIL_ROWS=0 &&
run "tail 1500" "" MA_LU_TAIL_ROWS=1500 &&
run "tail 2500" "" MA_LU_TAIL_ROWS=2500 &&
run "tail 3500" "" MA_LU_TAIL_ROWS=3500 &&
run "tail 4500" "" MA_LU_TAIL_ROWS=4500 &&
run "tail 5500" "" MA_LU_TAIL_ROWS=5500 &&
run "tail 3500 spacing 8" "" MA_LU_TAIL_ROWS=3500 MA_STAGE_SPACING=8 &&
run "tail 3500 spacing 10" "" MA_LU_TAIL_ROWS=3500 MA_STAGE_SPACING=10
