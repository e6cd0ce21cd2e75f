#!/bin/bash
# (record of an experiment: MA_FMM_NEAR_BLOCKS=2, MA_TMP_STRIP_WGS, MA_TMP_NEAR_U, MA_TMP_FMM_MASK existed only in the experimental builds described in profiles/r05_fmm_apply.md; tools/r05_fmm_strips_experiment.patch holds the strips kernel)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
L=gpurun_out/r05_fmm2.log; : > $L
timeout -k 10 600 python -m pytest tests/test_fmm_gpu.py tests/test_mlfmm_gpu.py tests/test_fmm_interface_gpu.py tests/test_box_gpu.py -x -q > gpurun_out/r05_tests_fmm.log 2>&1; echo "fmm tests: exit $?" | tee -a $L; tail -6 gpurun_out/r05_tests_fmm.log | tee -a $L
run() {
  echo "== $*" | tee -a $L
  env "$@" timeout -k 10 300 python tools/bench_mlfmm_box.py 1.0 64 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   mlfmm apply_ms %.4f near_GBs %.0f frac %.3f' % (d['apply_ms'], d['apply_near_GBs'], d['apply_near_GBs']/8000))" | tee -a $L
  env "$@" timeout -k 10 300 python tools/bench_slfmm_box.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   slfmm apply_ms %.4f near_GBs %.0f frac %.3f' % (d['apply_ms'], d['apply_near_GBs'], d['apply_near_GBs']/8000))" | tee -a $L
}
run MA_FMM_NEAR_BLOCKS=2
run MA_FMM_NEAR_BLOCKS=2 MA_TMP_STRIP_WGS=1
run MA_FMM_NEAR_BLOCKS=2 MA_TMP_STRIP_WGS=3
run MA_FMM_NEAR_BLOCKS=2 MA_FMM_OVERLAP=0
bash tools/r05_fmm_trace.sh strips2_seq MA_FMM_OVERLAP=0 > gpurun_out/r05_fmm_trace1.out 2>&1 && bash tools/r05_fmm_trace.sh strips2_ovl > gpurun_out/r05_fmm_trace2.out 2>&1
cat gpurun_out/r05_fmm_trace1.out gpurun_out/r05_fmm_trace2.out | cut -c1-120 | grep -v "at::native\|copyBuffer"
