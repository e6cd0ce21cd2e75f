#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab4.log; : > $L; : > gpurun_out/r03_ab4.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-timing $args 2>>gpurun_out/r03_ab4.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "reg mask40 3 slots" "" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 &&
run "reg mask40 4 slots" "--slots 4" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 MA_BENCH_MAX_SLOTS=8 &&
run "reg mask40 4 slots 8 queues" "--slots 4" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 MA_BENCH_MAX_SLOTS=8 GPU_MAX_HW_QUEUES=8 &&
run "reg mask40 5 slots 8 queues" "--slots 5 --steps 25 --warmup 5" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 MA_BENCH_MAX_SLOTS=8 GPU_MAX_HW_QUEUES=8 &&
run "reg mask40 6 slots 8 queues" "--slots 6 --warmup 6" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 MA_BENCH_MAX_SLOTS=8 GPU_MAX_HW_QUEUES=8 &&
run "reg mask64 6 slots 8 queues" "--slots 6 --warmup 6" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=64 MA_BENCH_MAX_SLOTS=8 GPU_MAX_HW_QUEUES=8 &&
run "reg nomask 4 slots 8 queues" "--slots 4" MA_LU_REG_PANEL=1 MA_BENCH_MAX_SLOTS=8 GPU_MAX_HW_QUEUES=8
