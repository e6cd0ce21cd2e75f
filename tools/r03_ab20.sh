#!/bin/bash
# few steps: which schedule should `auto` pick?
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab20.log; : > $L; : > gpurun_out/r03_ab20.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab20.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
for st in 3 6 9 12 20; do
  run "pipeline $st steps" "--steps $st --schedule pipeline" X=1 &&
  run "batch $st steps" "--steps $st --schedule batch" X=1 || exit 1
done
