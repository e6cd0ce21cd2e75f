#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab5.log; : > $L; : > gpurun_out/r03_ab5.err
MA_LIB_PATH=math_audio_amd/lib/libmathaudio_hip_stamps.so MA_LU_REG_PANEL=1 MA_LU_LOOKAHEAD=0 timeout -k 10 120 python tools/panel_reg_stamps.py 10000 2>&1 | grep -v amdgpu.ids | tee -a $L
MA_LU_REG_PANEL=1 timeout -k 10 600 python -m pytest tests/test_lu_gpu.py -q > gpurun_out/r03_lu_tests_reg.log 2>&1; echo "test_lu_gpu with MA_LU_REG_PANEL=1: exit $?" | tee -a $L; tail -5 gpurun_out/r03_lu_tests_reg.log | tee -a $L
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "old" "" MA_LU_REG_PANEL=0 &&
run "reg v4" "" MA_LU_REG_PANEL=1 &&
run "reg v4, big updates masked off 24 CUs" "" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=24 &&
run "reg v4, big updates masked off 40 CUs" "" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40 &&
run "reg v4, big updates masked off 48 CUs" "" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=48 &&
run "reg v4, big updates masked off 64 CUs" "" MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=64
