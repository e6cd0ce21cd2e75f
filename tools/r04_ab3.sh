#!/bin/bash
# round 4: the main lane's per-panel launches of a block as three launches (MA_LU_BLOCK_STEP), A/B on one box
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab3.log; : > $L; : > gpurun_out/r04_ab3.err
timeout -k 10 900 python -m pytest tests/test_lu_gpu.py -x -q > gpurun_out/r04_tests_lu_bs.log 2>&1; echo "lu tests (block step): exit $?" | tee -a $L; tail -15 gpurun_out/r04_tests_lu_bs.log | tee -a $L
timeout -k 10 900 python -m pytest tests/test_sweep_headline_gpu.py tests/test_sweep_gpu.py -x -q > gpurun_out/r04_tests_sweep_bs.log 2>&1; echo "sweep tests (block step): exit $?" | tee -a $L; tail -5 gpurun_out/r04_tests_sweep_bs.log | tee -a $L
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab3.err > gpurun_out/r04_ab3_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab3_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "block step on, 48" "--steps 48" X=1 &&
run "block step off, 48" "--steps 48" MA_LU_BLOCK_STEP=0 &&
run "block step on, no deferred finish" "--steps 48" MA_SWEEP_DEFER_FINISH=0 &&
run "block step on, split 56" "--steps 48" MA_LU_CU_SPLIT=56 &&
run "block step on, split 48" "--steps 48" MA_LU_CU_SPLIT=48 &&
run "block step on, kb 8" "--steps 48" MA_LU_KB=8 &&
run "block step on, 20" "--steps 20" X=1 &&
run "block step on, 48 again" "--steps 48" X=1
