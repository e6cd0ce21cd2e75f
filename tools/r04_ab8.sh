#!/bin/bash
# round 4 experiment: what would a panel kernel buy whose registers GUARANTEE two workgroups per CU (admission capacity doubled)?
# MA_LU_ADMIT_CUS=<c> makes the window count c CUs (one 253-register workgroup each): c = 2 x the panel CUs stands for two per CU.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab8.log; : > $L; : > gpurun_out/r04_ab8.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab8.err > gpurun_out/r04_ab8_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab8_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "default" "--steps 48" X=1
run "split 64, window counts 128" "--steps 48" MA_LU_ADMIT_CUS=128
run "split 64, window counts 96" "--steps 48" MA_LU_ADMIT_CUS=96
run "split 48, window counts 96" "--steps 48" MA_LU_CU_SPLIT=48 MA_LU_ADMIT_CUS=96
run "split 40, window counts 80" "--steps 48" MA_LU_CU_SPLIT=40 MA_LU_ADMIT_CUS=80
run "split 40, window counts 128" "--steps 48" MA_LU_CU_SPLIT=40 MA_LU_ADMIT_CUS=128
run "split 32, window counts 128" "--steps 48" MA_LU_CU_SPLIT=32 MA_LU_ADMIT_CUS=128
run "split 32, window counts 64" "--steps 48" MA_LU_CU_SPLIT=32 MA_LU_ADMIT_CUS=64
