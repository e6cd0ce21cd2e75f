#!/bin/bash
# round 3, first A/B on one box: register panel kernel and CU split against the round-2 default
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name, env...
  local name=$1; shift
  echo "== $name" | tee -a gpurun_out/r03_ab1.log
  env "$@" timeout -k 10 300 python bench.py --steps 24 --warmup 3 --no-cpu-baseline 2>>gpurun_out/r03_ab1.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f  roofline.frac %.3f  avg_launch_ms %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])); print('   phases', {k: round(v,2) for k,v in d['phase_ms_per_step'].items() if k != 'note'})" | tee -a gpurun_out/r03_ab1.log
}
: > gpurun_out/r03_ab1.log; : > gpurun_out/r03_ab1.err
MA_LU_REG_PANEL=1 timeout -k 10 600 python -m pytest tests/test_lu_gpu.py -x -q > gpurun_out/r03_lu_tests_reg.log 2>&1; echo "test_lu_gpu with MA_LU_REG_PANEL=1: exit $?" | tee -a gpurun_out/r03_ab1.log; tail -3 gpurun_out/r03_lu_tests_reg.log | tee -a gpurun_out/r03_ab1.log
run "round-2 default" MA_LU_REG_PANEL=0 &&
run "reg panel, no split" MA_LU_REG_PANEL=1 &&
run "reg panel, split 32" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32 &&
run "reg panel, split 40" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 &&
run "reg panel, split 24" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=24 &&
run "reg panel, split 32, chain masked" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32 MA_LU_CHAIN_MASK=1
