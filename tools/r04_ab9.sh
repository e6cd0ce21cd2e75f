#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
L=gpurun_out/r04_ab9.log; : > $L; : > gpurun_out/r04_ab9.err
timeout -k 10 600 python -m pytest tests/test_bem_assembly_gpu.py tests/test_bem_quad_gpu.py tests/test_sweep_headline_gpu.py tests/test_sweep_gpu.py -x -q > gpurun_out/r04_tests_near.log 2>&1; echo "assembly + sweep tests: exit $?" | tee -a $L; tail -3 gpurun_out/r04_tests_near.log | tee -a $L
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab9.err > gpurun_out/r04_ab9_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab9_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "near pairs of three systems per pass, 48" "--steps 48" X=1
run "a near launch per system, 48" "--steps 48" MA_BEM_NEAR_MULTI=0
run "near multi, 20" "--steps 20" X=1
run "near per system, 20" "--steps 20" MA_BEM_NEAR_MULTI=0
bash tools/r04_fmm5.sh > /dev/null 2>&1; cat gpurun_out/r04_fmm5.log | tee -a $L
