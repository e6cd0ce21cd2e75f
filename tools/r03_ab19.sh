#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab19.log; : > $L; : > gpurun_out/r03_ab19.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab19.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
timeout -k 10 300 python -m pytest tests/test_bem_assembly_gpu.py -q -x 2>&1 | tail -2 | tee -a $L
run "pieces: 4 per period" "" MA_BENCH_ASM_PIECES=4 &&
run "pieces: 2 per period" "" MA_BENCH_ASM_PIECES=2 &&
run "pieces: 3 per period" "" MA_BENCH_ASM_PIECES=3 &&
run "pieces: 6 per period" "" MA_BENCH_ASM_PIECES=6 &&
run "pieces: 9 per period (every round)" "" MA_BENCH_ASM_PIECES=9 &&
run "pieces: 1 per period" "" MA_BENCH_ASM_PIECES=1
