set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_sweep_gpu.py -x -q -m gpu > gpurun_out/r02_bp_tests.log 2>&1; echo rc=$? >> gpurun_out/r02_bp_tests.log; tail -4 gpurun_out/r02_bp_tests.log
run() {  # name, env..., -- bench args
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 24 --warmup 4 --schedule batch --no-cpu-baseline $EXTRA > gpurun_out/r02_bpx_$name.json 2> gpurun_out/r02_bpx_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_bpx_$name.json').read().strip().splitlines()[-1]); p=d['phase_ms_per_step']; print('$name', round(d['ms_per_step'],2), 'panel', round(p['lu_panel'],1), 'lanes_gemm', round(p['lu_zgemm_lookahead_lanes'],1), 'main_gemm', round(p['lu_zgemm'],1), 'trsm', round(p['lu_trsm'],1), 'swaps', round(p['lu_swaps'],1))
except Exception as e:
    print('$name failed', e); print(open('gpurun_out/r02_bpx_$name.err').read()[-400:])
PY
}
EXTRA="--slots 3" run lds74 MA_LU_BATCH_LDS=74
EXTRA="--slots 3" run lds110 MA_LU_BATCH_LDS=110
EXTRA="--slots 3" run lds56 MA_LU_BATCH_LDS=56
EXTRA="--slots 4" run s4lds98 MA_LU_BATCH_LDS=98
EXTRA="--slots 2" run s2lds50 MA_LU_BATCH_LDS=50
EXTRA="--slots 3" run nb16 MA_LU_BATCH_NB=16 MA_LU_BATCH_LDS=40
