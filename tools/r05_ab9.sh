#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab9.log; : > $L; : > gpurun_out/r05_ab9.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --warmup 3 --no-cpu-baseline --no-extras --no-check $args 2>>gpurun_out/r05_ab9.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "split 32, 48 steps, no timing events" "--steps 48 --no-timing" MA_LU_CU_SPLIT=32 &&
run "split 32, 48 steps, timing" "--steps 48" MA_LU_CU_SPLIT=32 &&
run "split 32, 96 steps, no timing" "--steps 96 --no-timing" MA_LU_CU_SPLIT=32 &&
run "split 32, 20 steps, no timing" "--steps 20 --no-timing" MA_LU_CU_SPLIT=32 &&
run "split 64, 48 steps, no timing" "--steps 48 --no-timing" MA_LU_CU_SPLIT=64 &&
run "split 0, 48 steps, no timing" "--steps 48 --no-timing" MA_LU_CU_SPLIT=0 &&
run "split 32, 48 steps, no timing, verified" "--steps 48 --no-timing" MA_LU_CU_SPLIT=32 MA_SWEEP_SPECULATE=verified &&
run "split 32, 48 steps, no timing, partial verified" "--steps 48 --no-timing" MA_LU_CU_SPLIT=32 MA_SWEEP_SPECULATE=verified MA_SWEEP_PIVOTING=partial
