#!/bin/bash
# below the default's lower bound: does the pair + split pay at 2 000 / 4 000 rows?
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab13.log; : > $L; : > gpurun_out/r03_ab13.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab13.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.3f' % d['ms_per_step'])" | tee -a $L
}
for nt in 11 21; do
  run "old n_theta $nt" "--n-theta $nt --steps 48" MA_LU_REG_PANEL=0 MA_LU_CU_SPLIT=0 &&
  run "pair 64 n_theta $nt" "--n-theta $nt --steps 48" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 &&
  run "pair 32 n_theta $nt" "--n-theta $nt --steps 48" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=32 &&
  run "pair no split n_theta $nt" "--n-theta $nt --steps 48" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=0 || exit 1
done
