#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03prof3; rm -rf $O; mkdir -p $O
for cfg in "reg_mask40 MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=40"; do
  set -- $cfg; name=$1; shift
  env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/$name -- python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-timing --no-extras > $O/$name.json 2> $O/$name.err || exit 1
  t=$(find $O/$name -name "*kernel_trace.csv" | head -1)
  python tools/chain_analysis.py $t $O/${name}_chain.json > /dev/null 2>&1
  python - "$t" > $O/${name}_window.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Queue_Id"],r["Kernel_Name"].split("(")[0].replace("void ","").replace("ma::","")[:28],int(r.get("Grid_Size_X",r.get("Grid_Size",0)) or 0)) for r in rows)
T0=ev[0][0]; T1=max(e[1] for e in ev)
mid=T0+(T1-T0)*0.5
win=[e for e in ev if e[0]>=mid and e[0]<mid+12e6]
qs=sorted(set(e[2] for e in win))
print("queues",qs)
for e in win:
    print("%9.1f us dur %8.1f q%-3s %-28s grid %d" % ((e[0]-mid)/1e3,(e[1]-e[0])/1e3,e[2],e[3],e[4]))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03prof3/reg_mask40_chain.json"))
print(json.load(open("gpurun_out/r03prof3/reg_mask40.json"))["ms_per_step"])
for kk,vv in list(d["per_kernel"].items())[:9]: print("   ",kk, {a:round(b,2) for a,b in vv.items()})
print(d["chip"]); print(d["zgemm_concurrency_frac"])
for q,v in d["streams"].items(): print(q, {k:(round(x,2) if isinstance(x,float) else x) for k,x in v.items() if k not in ("kinds","chain")}, v.get("chain"))
for k,v in d["gaps_over_100us"].items(): print("gap",k,v)
PY
