#!/bin/bash
# round 4: (d) sensitivity of the step to the panel's time per column (diagnostic sleep), and kb at the driver's 20 steps
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_ab5.log; : > $L; : > gpurun_out/r04_ab5.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args 2>>gpurun_out/r04_ab5.err > gpurun_out/r04_ab5_last.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_ab5_last.json')); print('   ms_per_step %.2f  check %s  phases %s' % (d['ms_per_step'], d.get('check',{}).get('max_rel_residual'), {k: round(v,2) for k,v in d.get('phase_ms_per_step',{}).items() if isinstance(v,float)}))" | tee -a $L
}
run "default 48" "--steps 48" X=1
run "panel + 1 x 0.21 us per column" "--steps 48" MA_DIAG_PANEL_SLEEP=1
run "panel + 3 x 0.21 us per column" "--steps 48" MA_DIAG_PANEL_SLEEP=3
run "panel + 5 x 0.21 us per column" "--steps 48" MA_DIAG_PANEL_SLEEP=5
run "panel + 10 x 0.21 us per column" "--steps 48" MA_DIAG_PANEL_SLEEP=10
run "default 20" "--steps 20" X=1
run "kb 7, 20" "--steps 20" MA_LU_KB=7
run "kb 8, 20" "--steps 20" MA_LU_KB=8
run "kb 8, 48" "--steps 48" MA_LU_KB=8
run "default 20 again" "--steps 20" X=1
