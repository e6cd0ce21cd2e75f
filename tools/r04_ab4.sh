#!/bin/bash
# round 4 diagnostic: what do the lanes' small update launches cost the step? (results are WRONG with the skip switch: timing only)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab4.log; : > $L; : > gpurun_out/r04_ab4.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-check $args 2>>gpurun_out/r04_ab4.err > gpurun_out/r04_ab4_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab4_last.json')); print('   ms_per_step %.2f  phases %s' % (d['ms_per_step'], {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "default" "--steps 48" X=1
run "skip lane gemms K<=32" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=32
run "skip lane gemms K<=64" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=64
run "skip lane gemms K<=64, block step off" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=64 MA_LU_BLOCK_STEP=0
run "skip lane gemms K<=384 (narrow too)" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=384
run "skip K<=64, split 48" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=64 MA_LU_CU_SPLIT=48
run "skip K<=64, split 40" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=64 MA_LU_CU_SPLIT=40
run "skip K<=64, split 32" "--steps 48" MA_DIAG_SKIP_LANE_GEMM=64 MA_LU_CU_SPLIT=32
