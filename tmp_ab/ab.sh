set -e
L=math_audio_amd/lib/libmathaudio_hip.so
cp tmp_ab/new.so $L
timeout -k 10 400 python -m pytest tests/test_lu_gpu.py -m gpu -x -q > gpurun_out/ab_tests.log 2>&1 || { tail -20 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for i in 1 2; do
  cp tmp_ab/base.so $L && timeout -k 10 200 python bench.py --steps 24 --warmup 3 > gpurun_out/ab_base_$i.json 2>gpurun_out/ab_err.log
  cp tmp_ab/new.so $L && timeout -k 10 200 python bench.py --steps 24 --warmup 3 > gpurun_out/ab_new_$i.json 2>gpurun_out/ab_err.log
done
python - <<'PY'
import json
for k in ("base_1","new_1","base_2","new_2"):
    d=json.loads(open("gpurun_out/ab_%s.json"%k).read().strip().splitlines()[-1]); print(k, d["ms_per_step"])
PY
