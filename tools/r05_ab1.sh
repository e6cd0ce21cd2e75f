#!/bin/bash
# round 5, item 1: tournament-pivot panels -- correctness first, then the CU split re-swept (one box)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab1.log; : > $L; : > gpurun_out/r05_ab1.err
timeout -k 10 900 python -m pytest tests/test_lu_tournament_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee -a $L || exit 1
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 200 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-extras $args 2>>gpurun_out/r05_ab1.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f  phases %s  check %s' % (d['ms_per_step'], d.get('phase_ms_per_step'), d.get('check')))" | tee -a $L
}
run "partial (round-4 default)" "" MA_SWEEP_PIVOTING=partial &&
run "tournament split 64" "" X=1 &&
run "tournament split 48" "" MA_LU_CU_SPLIT=48 &&
run "tournament split 32" "" MA_LU_CU_SPLIT=32 &&
run "tournament split 16" "" MA_LU_CU_SPLIT=16 &&
run "tournament split 0" "" MA_LU_CU_SPLIT=0 &&
run "tournament split 32 slots 2" "--slots 2" MA_LU_CU_SPLIT=32 &&
run "tournament split 0 slots 2" "--slots 2" MA_LU_CU_SPLIT=0 &&
run "tournament split 0 slots 4" "--slots 4" MA_LU_CU_SPLIT=0
