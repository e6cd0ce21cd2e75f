#!/bin/bash
# round 4 experiment: six systems on the three lane streams (MA_LU_LANE_ALIAS=3: slots m and m + 3 share a lane), A/B on one box
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab10.log; : > $L; : > gpurun_out/r04_ab10.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab10.err > gpurun_out/r04_ab10_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab10_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "default (3 slots, 3 lanes)" "--steps 48" X=1
run "6 slots on 3 lanes" "--steps 48 --slots 6" MA_LU_LANE_ALIAS=3
run "5 slots on 3 lanes" "--steps 48 --slots 5" MA_LU_LANE_ALIAS=3
run "4 slots on 3 lanes" "--steps 48 --slots 4" MA_LU_LANE_ALIAS=3
run "4 slots on 2 lanes" "--steps 48 --slots 4" MA_LU_LANE_ALIAS=2
run "6 slots on 3 lanes, 20 steps" "--steps 20 --slots 6" MA_LU_LANE_ALIAS=3
run "6 slots on 3 lanes, split 48" "--steps 48 --slots 6" MA_LU_LANE_ALIAS=3 MA_LU_CU_SPLIT=48
