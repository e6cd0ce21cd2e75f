#!/bin/bash
# round 3: kernel trace of the sweep with the register panel kernel and the big updates masked off P CUs
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03prof2; rm -rf $O; mkdir -p $O
for cfg in "old MA_LU_REG_PANEL=0" "reg_mask40 MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40"; do
  set -- $cfg; name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-timing > $O/$name.json 2> $O/$name.err || exit 1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); cut -c1-200 $f > $O/${name}_kernel_stats.csv
  t=$(find $O/$name -name "*kernel_trace.csv" | head -1)
  python tools/chain_analysis.py $t $O/${name}_chain.json > /dev/null 2>&1
  python tools/kernel_timeline.py $t 3000 > $O/${name}_timeline.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
for name in ("old","reg_mask40"):
    d=json.load(open("gpurun_out/r03prof2/%s_chain.json"%name))
    print("==",name, "ms/freq in window", round(d["ms_per_frequency_in_window"],1), json.load(open("gpurun_out/r03prof2/%s.json"%name))["ms_per_step"])
    for kk,vv in list(d["per_kernel"].items())[:8]: print("   ",kk, {a:round(b,2) for a,b in vv.items()})
    print("   chip", {k:round(v,3) for k,v in d["chip"].items()})
    for k,v in d["panel_us_by_company"].items(): print("   panel", k, v["n"], round(v["avg_us"],1))
PY
