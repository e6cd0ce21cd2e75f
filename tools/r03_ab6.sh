#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r03_ab6.log; : > $L; : > gpurun_out/r03_ab6.err
MA_LU_REG_PANEL=2 timeout -k 10 600 python -m pytest tests/test_lu_gpu.py -q > gpurun_out/r03_lu_tests_pair.log 2>&1; echo "test_lu_gpu with MA_LU_REG_PANEL=2: exit $?" | tee -a $L; tail -4 gpurun_out/r03_lu_tests_pair.log | tee -a $L
run() { local name=$1; shift; local args=$1; shift
  echo "== $name ($args)" | tee -a $L
  env "$@" timeout -k 10 300 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-timing --no-extras $args 2>>gpurun_out/r03_ab6.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
}
run "old" "" MA_LU_REG_PANEL=0 &&
run "pair" "" MA_LU_REG_PANEL=2 &&
run "pair, big updates masked off 32" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=32 &&
run "pair, big updates masked off 40" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=40 &&
run "pair, big updates masked off 48" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=48 &&
run "pair, big updates masked off 64" "" MA_LU_REG_PANEL=2 MA_LU_CU_SPLIT=64
O=gpurun_out/r03prof_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 12 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 12 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "python3 bench.py --steps 12 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras"
rm -rf $O/pmc_fetch $O/pmc_write
