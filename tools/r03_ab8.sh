#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab8.log; : > $L; : > gpurun_out/r03_ab8.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab8.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "old" "" MA_LU_REG_PANEL=0 &&
run "pair mask 64" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 &&
run "pair mask 64 asm on lanes" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 MA_BENCH_ASM_LANE=1 &&
run "pair mask 56 asm on lanes" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=56 MA_BENCH_ASM_LANE=1 &&
run "pair mask 48 asm on lanes" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=48 MA_BENCH_ASM_LANE=1 &&
run "pair mask 64 spacing 13 asm on lanes" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 MA_STAGE_SPACING=13 MA_BENCH_ASM_LANE=1 &&
run "pair mask 64 model order" "--schedule model" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64
