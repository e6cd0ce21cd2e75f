#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab2.log; : > $L
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 24 --warmup 3 --no-cpu-baseline $args 2>>gpurun_out/r03_ab2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
: > gpurun_out/r03_ab2.err
run "old" "--no-timing" MA_LU_REG_PANEL=0 &&
run "reg" "--no-timing" MA_LU_REG_PANEL=1 &&
run "reg split32" "--no-timing" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32 &&
run "reg split32" "" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32 &&
run "reg split32 chainmask" "--no-timing" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32 MA_LU_CHAIN_MASK=1 &&
run "reg split32 48 steps" "--no-timing --steps 48" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32 &&
run "old 48 steps" "--no-timing --steps 48" MA_LU_REG_PANEL=0
