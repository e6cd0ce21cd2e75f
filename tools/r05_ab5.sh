#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab5.log; : > $L; : > gpurun_out/r05_ab5.err
timeout -k 10 900 python -m pytest tests/test_lu_tournament_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee -a $L
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-extras $args 2>>gpurun_out/r05_ab5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('phase_ms_per_step') or {}; q=d['lu_panels']; print('   ms_per_step %.2f  big %.2f asm %.2f neither %.2f lane %.2f  res %.2e  %s %s acc %.1f wid %.1f rej %.1f' % (d['ms_per_step'], p.get('big_updates',0), p.get('assembly_in_the_timed_region',0), p.get('stream_neither',0), p.get('lane_updates',0), d['check']['max_rel_residual'], q['pivoting'], q['speculation'], q['half_panels_accepted_per_step'], q['half_panels_accepted_widened_per_step'], q['half_panels_rejected_per_step']))" | tee -a $L
}
run "optimistic, split 64" "" X=1 &&
run "optimistic, split 32" "" MA_LU_CU_SPLIT=32 &&
run "verified, split 32" "" MA_SWEEP_SPECULATE=verified MA_LU_CU_SPLIT=32 &&





run "optimistic, split 32, 20 steps" "--steps 20" MA_LU_CU_SPLIT=32
