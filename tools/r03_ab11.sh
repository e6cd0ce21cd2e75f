#!/bin/bash
# pair mode + CU split over system sizes: where does it pay?
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab11.log; : > $L; : > gpurun_out/r03_ab11.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab11.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
for nt in 31 41 61 71; do
  st=36; [ $nt -ge 61 ] && st=24
  run "old n_theta $nt" "--n-theta $nt --steps $st" MA_LU_REG_PANEL=0 &&
  run "pair 64 n_theta $nt" "--n-theta $nt --steps $st" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 || exit 1
done
run "pair 48 n_theta 31" "--n-theta 31 --steps 36" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=48 &&
run "pair 80 n_theta 71" "--n-theta 71 --steps 24" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=80
