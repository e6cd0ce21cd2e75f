#!/bin/bash
# round 4: the sweep handle, its new tests, and the deferred backward substitution (A/B on one box)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab2.log; : > $L; : > gpurun_out/r04_ab2.err
timeout -k 10 900 python -m pytest tests/test_sweep_headline_gpu.py tests/test_sweep_gpu.py tests/test_residency_gpu.py -x -q > gpurun_out/r04_tests_sweep.log 2>&1; echo "sweep tests: exit $?" | tee -a $L; tail -5 gpurun_out/r04_tests_sweep.log | tee -a $L
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab2.err > gpurun_out/r04_ab2_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab2_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "default 48" "--steps 48" X=1 &&
run "no deferred finish 48" "--steps 48" MA_SWEEP_DEFER_FINISH=0 &&
run "defer block 0, 48" "--steps 48" MA_SWEEP_DEFER_BLOCK=0 &&
run "defer block 2, 48" "--steps 48" MA_SWEEP_DEFER_BLOCK=2 &&
run "default 20" "--steps 20" X=1 &&
run "no deferred finish 20" "--steps 20" MA_SWEEP_DEFER_FINISH=0 &&
run "default 48 no timing" "--steps 48 --no-timing" X=1 &&
run "default 48 again" "--steps 48" X=1
cp gpurun_out/r04_ab2_last.json gpurun_out/r04_bench_default48.json
timeout -k 10 1000 python -m pytest tests/test_lu_gpu.py -x -q > gpurun_out/r04_tests_lu.log 2>&1; echo "lu tests: exit $?" | tee -a $L; tail -3 gpurun_out/r04_tests_lu.log | tee -a $L
