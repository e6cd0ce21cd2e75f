# A/B runs of bench.py on one box: tools/ab.sh reads lines "name ENV=.. ENV=.. -- extra bench args" from tools/ab_cases.txt
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
while IFS= read -r line; do
  [ -z "$line" ] && continue
  name=${line%% *}; rest=${line#* }
  envs=${rest%%--*}; extra=${rest#*--}
  [ "$extra" = "$rest" ] && extra=""
  env $envs timeout -k 10 240 python bench.py --steps ${AB_STEPS:-24} --warmup 6 --no-cpu-baseline $extra > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1]); p=d['phase_ms_per_step']; print('$name', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'main_gemm', round(p['lu_zgemm'],1), 'lanes_gemm', round(p['lu_zgemm_lookahead_lanes'],1), flush=True)
except Exception as e:
    print('$name failed', e); print(open('gpurun_out/ab/$name.err').read()[-600:])
PY
done < tools/ab_cases.txt
