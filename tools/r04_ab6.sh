#!/bin/bash
# round 4: the assembly-ahead on a stream of its own (masked to the panel CUs / unmasked), A/B on one box
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab6.log; : > $L; : > gpurun_out/r04_ab6.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab6.err > gpurun_out/r04_ab6_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab6_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "default 48" "--steps 48" X=1
run "assembly on a stream masked to the 64 panel CUs" "--steps 48" MA_SWEEP_ASM_STREAM=1
run "assembly on an unmasked stream of its own" "--steps 48" MA_SWEEP_ASM_STREAM=2
run "masked asm stream, 6 pieces per system" "--steps 48" MA_SWEEP_ASM_STREAM=1 MA_SWEEP_ASM_PIECES=8
run "masked asm stream, split 56" "--steps 48" MA_SWEEP_ASM_STREAM=1 MA_LU_CU_SPLIT=56
run "masked asm stream, 20 steps" "--steps 20" MA_SWEEP_ASM_STREAM=1
MA_SWEEP_ASM_STREAM=1 timeout -k 10 600 python -m pytest tests/test_sweep_headline_gpu.py -x -q > gpurun_out/r04_tests_asmstream.log 2>&1; echo "headline tests with the assembly stream: exit $?" | tee -a $L; tail -3 gpurun_out/r04_tests_asmstream.log | tee -a $L
