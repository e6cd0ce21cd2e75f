set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_lu_gpu.py -x -q -m gpu -k "staged or batched or abandoned" > gpurun_out/r02_bp_tests.log 2>&1; echo rc=$? >> gpurun_out/r02_bp_tests.log; tail -6 gpurun_out/r02_bp_tests.log
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline $EXTRA > gpurun_out/r02_bpg_$name.json 2> gpurun_out/r02_bpg_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_bpg_$name.json').read().strip().splitlines()[-1]); p=d['phase_ms_per_step']; print('$name', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'main_gemm', round(p['lu_zgemm'],1), 'lanes_gemm', round(p['lu_zgemm_lookahead_lanes'],1))
except Exception as e:
    print('$name failed', e); print(open('gpurun_out/r02_bpg_$name.err').read()[-500:])
PY
}
EXTRA="--slots 3" run base_pipeline A=1
EXTRA="--slots 6 --group-size 3" run g3x2 A=1
EXTRA="--slots 6 --group-size 3" run g3x2_lds74 MA_LU_BATCH_LDS=74
EXTRA="--slots 6 --group-size 2" run g2x3 A=1
EXTRA="--slots 8 --group-size 4" run g4x2 MA_LU_BATCH_LDS=76
EXTRA="--slots 4 --group-size 2" run g2x2 A=1
