#!/bin/bash
# round 5: wide near blocks on the matrix cores (slfmm_near_mfma_kernel, MA_FMM_NEAR_BLOCKS=2 = default) against the vector block kernels (=1)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
L=gpurun_out/r05_fmm4.log; : > $L
timeout -k 10 900 python -m pytest tests/test_fmm_gpu.py tests/test_mlfmm_gpu.py tests/test_fmm_interface_gpu.py tests/test_box_gpu.py -x -q > gpurun_out/r05_tests_fmm.log 2>&1; echo "fmm tests: exit $?" | tee -a $L; tail -6 gpurun_out/r05_tests_fmm.log | tee -a $L
run() {
  echo "== $*" | tee -a $L
  env "$@" timeout -k 10 300 python tools/bench_slfmm_box.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   slfmm apply_ms %.4f near_GBs %.0f frac %.3f' % (d['apply_ms'], d['apply_near_GBs'], d['apply_near_GBs']/8000))" | tee -a $L
}
run MA_FMM_NEAR_BLOCKS=1
run MA_FMM_NEAR_BLOCKS=2
run MA_FMM_NEAR_BLOCKS=2 MA_FMM_OVERLAP=0
run MA_FMM_NEAR_BLOCKS=1 MA_FMM_OVERLAP=0
O=gpurun_out/r05_fmm4_trace; rm -rf $O; mkdir -p $O
for mode in 0 1; do
rocprofv3 --kernel-trace --output-format csv -d $O/sl$mode -- python3 tools/bench_slfmm_box.py > $O/sl$mode.json 2> $O/sl$mode.err
f=$(find $O/sl$mode -name "*kernel_trace.csv" | head -1)
MA_DUMMY=$mode python3 - "$f" <<'PY' | tee -a $L
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"]]
i0 = idx[-1]; t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 9]:
    print("%9.1f %9.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
PY
rm -rf $O/sl$mode
export MA_FMM_OVERLAP=0
done
