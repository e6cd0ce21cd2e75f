#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab7.log; : > $L; : > gpurun_out/r04_ab7.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab7.err > gpurun_out/r04_ab7_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab7_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "first system alone, 20" "--steps 20" X=1
run "first three together, 20" "--steps 20" MA_SWEEP_FIRST_ALONE=0
run "first system alone, 48" "--steps 48" X=1
run "first three together, 48" "--steps 48" MA_SWEEP_FIRST_ALONE=0
run "first system alone, 20 again" "--steps 20" X=1
run "first three together, 20 again" "--steps 20" MA_SWEEP_FIRST_ALONE=0
run "first alone, 8 pieces per begin, 20" "--steps 20" MA_SWEEP_ASM_PIECES=8
timeout -k 10 600 python -m pytest tests/test_sweep_headline_gpu.py tests/test_sweep_gpu.py -x -q > gpurun_out/r04_tests_firstalone.log 2>&1; echo "sweep tests: exit $?" | tee -a $L; tail -3 gpurun_out/r04_tests_firstalone.log | tee -a $L
