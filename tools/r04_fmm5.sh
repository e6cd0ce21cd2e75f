#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
L=gpurun_out/r04_fmm5.log; : > $L
timeout -k 10 600 python -m pytest tests/test_fmm_gpu.py tests/test_mlfmm_gpu.py tests/test_fmm_interface_gpu.py tests/test_box_gpu.py -x -q > gpurun_out/r04_tests_fmm.log 2>&1; echo "fmm tests: exit $?" | tee -a $L; tail -3 gpurun_out/r04_tests_fmm.log | tee -a $L
ml() { echo "== mlfmm $*" | tee -a $L; env "$@" timeout -k 10 300 python tools/bench_mlfmm_box.py 1.0 64 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   apply_ms %.4f frac %.3f' % (d['apply_ms'], d['apply_near_GBs']/8000))" | tee -a $L; }
ml X=1
ml MA_FMM_OVERLAP=0
bash tools/r04_fmm_trace.sh > /dev/null 2>&1; grep -E "near_blocks|near_gather" gpurun_out/r04_fmm_trace.txt | tail -3 | tee -a $L
