#!/bin/bash
# (historical: MA_TEST_STAGE_SPACING was a hook of the diagnostic build while the spacing was being measured; it has been removed since)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_ab12.log; : > $L
timeout -k 10 900 python -m pytest tests/test_config2_gpu.py tests/test_sharded_gpu.py tests/test_sweep_gpu.py tests/test_operator_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee -a $L
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench20.json 2> gpurun_out/r05_bench20.err || { tail -5 gpurun_out/r05_bench20.err; exit 1; }
python - <<'PY' | tee -a $L
import json
d = json.load(open("gpurun_out/r05_bench20.json"))
print("ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac")}, "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("blas_threads"))
print("proxy", json.dumps(d.get("strong_scaling_proxy"), indent=None)[:900])
print("lu_panels", {k: v for k, v in d["lu_panels"].items() if k != "note"})
PY
for sp in 9 7 6 5; do
  echo "== K = 8, spacing $sp (diagnostic library)" | tee -a $L
  MA_LIB_PATH=$PWD/math_audio_amd/lib/libmathaudio_hip_diag.so MA_TEST_STAGE_SPACING=$sp timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-check --no-timing | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
done
for sp in 9 7; do
  echo "== K = 64, spacing $sp (diagnostic library)" | tee -a $L
  MA_LIB_PATH=$PWD/math_audio_amd/lib/libmathaudio_hip_diag.so MA_TEST_STAGE_SPACING=$sp timeout -k 10 200 python bench.py --steps 64 --warmup 3 --no-cpu-baseline --no-extras --no-check --no-timing | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
done
