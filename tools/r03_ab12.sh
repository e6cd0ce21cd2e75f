#!/bin/bash
# the one-shot paths (no staged pipeline): does the register pair panel / the CU split pay there?
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab12.log; : > $L; : > gpurun_out/r03_ab12.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --warmup 2 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab12.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "single old" "--schedule batch --slots 1 --steps 10" MA_LU_REG_PANEL=0 &&
run "single pair" "--schedule batch --slots 1 --steps 10" MA_LU_REG_PANEL=2 &&
run "single pair split 64" "--schedule batch --slots 1 --steps 10" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64 &&
run "single reg32" "--schedule batch --slots 1 --steps 10" MA_LU_REG_PANEL=1 &&
run "batch3 old" "--schedule batch --slots 3 --steps 12" MA_LU_REG_PANEL=0 &&
run "batch3 pair" "--schedule batch --slots 3 --steps 12" MA_LU_REG_PANEL=2 &&
run "batch3 pair split 64" "--schedule batch --slots 3 --steps 12" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64
