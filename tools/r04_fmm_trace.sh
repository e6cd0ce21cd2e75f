#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r04_fmm_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/ml -- python3 tools/bench_mlfmm_box.py 1.0 64 1000 > $O/ml.json 2> $O/ml.err
f=$(find $O/ml -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/r04_fmm_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last apply: find the last fillBuffer (memset of y) and print everything after it
idx = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    print("%9.1f %9.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
rm -rf $O/ml
tail -40 gpurun_out/r04_fmm_trace.txt
