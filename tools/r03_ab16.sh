#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
L=gpurun_out/r03_ab16.log; : > $L; : > gpurun_out/r03_ab16.err
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab16.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "default" "" X=1 &&
run "split as 2 whole XCDs" "" MA_LU_SPLIT_SHAPE=xcd &&
run "lanes high priority" "" MA_LU_LANE_PRIO=1 &&
run "lanes low priority" "" MA_LU_LANE_PRIO=-1 &&
run "split 56" "" MA_LU_CU_SPLIT=56 &&
run "asm ahead 2" "" MA_BENCH_ASM_AHEAD=2 &&
run "default again" "" X=1
