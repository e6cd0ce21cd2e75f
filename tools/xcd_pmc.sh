cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/xcdpmc; mkdir -p $O
MA_ZGEMM_XCD_TILES=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/off -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-timing --schedule batch > /dev/null 2> $O/off.err && echo off ok
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/one -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-timing --schedule batch > /dev/null 2> $O/one.err && echo one ok
MA_ZGEMM_XCD_PERSIST=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/persist -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-timing --schedule batch > /dev/null 2> $O/persist.err && echo persist ok
python - <<'PY'
import csv,glob,collections
for mode in ("off","one","persist"):
    acc=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob("gpurun_out/xcdpmc/%s/**/*counter_collection.csv"%mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name")!="FETCH_SIZE": continue
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            acc[k][0]+=1; acc[k][1]+=float(r["Counter_Value"])
    for k,v in acc.items():
        if "zgemm3m" in k or "lu_panel" in k: print(mode,k,v[0],"launches, fetch MB per launch (x2 corrected):", round(2*v[1]/v[0]/1024,1), "total GB:", round(2*v[1]/1024/1024,2))
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
