#!/bin/bash
# round 5: kernel timeline of the last apply of the multi-level and single-level operators of config #5 (arguments: NAME ENV=VAL ...)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
name=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/r05_fmm_trace_$name; rm -rf $O; mkdir -p $O
for which in ml sl; do
  if [ $which = ml ]; then rocprofv3 --kernel-trace --output-format csv -d $O/$which -- python3 tools/bench_mlfmm_box.py 1.0 64 1000 > $O/$which.json 2> $O/$which.err
  else rocprofv3 --kernel-trace --output-format csv -d $O/$which -- python3 tools/bench_slfmm_box.py > $O/$which.json 2> $O/$which.err; fi
  f=$(find $O/$which -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY' > gpurun_out/r05_fmm_trace_${name}_$which.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "slfmm_up_" in r["Kernel_Name"]]    # an apply starts with the upward pass (round 5: no memset any more)
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 30]:
    print("%9.1f %9.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
PY
  rm -rf $O/$which
  echo "---- $name $which"; head -24 gpurun_out/r05_fmm_trace_${name}_$which.txt
done
