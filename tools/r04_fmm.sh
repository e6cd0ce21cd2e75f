#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
L=gpurun_out/r04_fmm.log; : > $L
timeout -k 10 600 python -m pytest tests/test_fmm_gpu.py tests/test_mlfmm_gpu.py tests/test_fmm_interface_gpu.py -x -q > gpurun_out/r04_tests_fmm.log 2>&1; echo "fmm tests: exit $?" | tee -a $L; tail -6 gpurun_out/r04_tests_fmm.log | tee -a $L
for ov in 1 0; do for bl in 1; do
  echo "== MA_FMM_OVERLAP=$ov MA_FMM_BATCH_LEVELS=$bl" | tee -a $L
  MA_FMM_OVERLAP=$ov MA_FMM_BATCH_LEVELS=$bl timeout -k 10 300 python tools/bench_mlfmm_box.py 1.0 64 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   mlfmm apply_ms %.4f near_GBs %.0f frac %.3f' % (d['apply_ms'], d['apply_near_GBs'], d['apply_near_GBs']/8000))" | tee -a $L
done; done
for ov in 1 0; do
  echo "== slfmm MA_FMM_OVERLAP=$ov" | tee -a $L
  MA_FMM_OVERLAP=$ov timeout -k 10 300 python tools/bench_slfmm_box.py 2>/dev/null | tail -1 | cut -c1-600 | tee -a $L
done
