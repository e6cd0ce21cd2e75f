#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab10.log; : > $L; : > gpurun_out/r03_ab10.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab10.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "pair mask 64" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 &&
run "pair mask 64 admit 64" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 MA_LU_ADMIT_CUS=64 &&
run "pair mask 48 admit 48" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=48 MA_LU_ADMIT_CUS=48 &&
run "pair mask 40 admit 40" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=40 MA_LU_ADMIT_CUS=40 &&
run "pair mask 32 admit 32" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=32 MA_LU_ADMIT_CUS=32 &&
run "pair mask 48 admit 40" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=48 MA_LU_ADMIT_CUS=40 &&
run "pair mask 56 admit 40" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=56 MA_LU_ADMIT_CUS=40
