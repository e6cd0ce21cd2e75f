set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_lu_gpu.py tests/test_sweep_gpu.py -x -q -m gpu > gpurun_out/r02_bp_tests.log 2>&1; echo rc=$? >> gpurun_out/r02_bp_tests.log; tail -12 gpurun_out/r02_bp_tests.log
for cfg in "pipeline 1" "batch 0" "batch 1"; do
  set -- $cfg
  MA_LU_BATCH_PANEL=$2 timeout -k 10 200 python bench.py --steps 24 --warmup 3 --schedule $1 --no-cpu-baseline > gpurun_out/r02_bp_$1_$2.json 2> gpurun_out/r02_bp_$1_$2.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_bp_$1_$2.json').read().strip().splitlines()[-1]); print('$1 bp=$2', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), d['phase_ms_per_step']['lu_panel'])
except Exception as e:
    print('$1 bp=$2 failed', e); print(open('gpurun_out/r02_bp_$1_$2.err').read()[-600:])
PY
done
