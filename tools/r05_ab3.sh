#!/bin/bash
# round 5, item 1: the speculative panel -- correctness first, then the CU split x slots re-swept (one box)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab3.log; : > $L; : > gpurun_out/r05_ab3.err
timeout -k 10 900 python -m pytest tests/test_lu_tournament_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee -a $L
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 200 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-extras $args 2>>gpurun_out/r05_ab3.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('phase_ms_per_step') or {}; print('   ms_per_step %.2f  big %.2f asm %.2f neither %.2f lane %.2f  res %.2e' % (d['ms_per_step'], p.get('big_updates',0), p.get('assembly_in_the_timed_region',0), p.get('stream_neither',0), p.get('lane_updates',0), d['check']['max_rel_residual']))" | tee -a $L
}
run "partial, no speculation (round 4)" "" MA_SWEEP_PIVOTING=partial MA_LU_SPECULATE=0 MA_SWEEP_SPECULATE=off &&
run "tournament, verified speculation, split 32" "" MA_SWEEP_SPECULATE=verified MA_LU_CU_SPLIT=32 &&
run "partial + speculation, split 64" "" MA_SWEEP_PIVOTING=partial &&
run "tournament + speculation, split 64" "" X=1 &&
run "tournament + speculation, split 48" "" MA_LU_CU_SPLIT=48 &&
run "tournament + speculation, split 32" "" MA_LU_CU_SPLIT=32 &&
run "tournament + speculation, split 24" "" MA_LU_CU_SPLIT=24 &&
run "tournament + speculation, split 16" "" MA_LU_CU_SPLIT=16 &&
run "tournament + speculation, split 0" "" MA_LU_CU_SPLIT=0 &&
run "tournament + speculation, split 32 kb 8" "" MA_LU_CU_SPLIT=32 MA_LU_KB=8 &&
run "tournament + speculation, split 32 kb 4" "" MA_LU_CU_SPLIT=32 MA_LU_KB=4 &&
run "tournament + speculation, split 32 slots 4" "--slots 4" MA_LU_CU_SPLIT=32 &&
run "tournament + speculation, split 32 slots 2" "--slots 2" MA_LU_CU_SPLIT=32
