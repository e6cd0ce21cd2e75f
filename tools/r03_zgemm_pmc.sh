#!/bin/bash
# where do the update kernel's wavefronts spend their cycles? (SQ counters, one pass each set)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_zgemm_pmc; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/p1 -o p1 -- python3 tools/zgemm_one.py 9984 9984 1024 3 > $O/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/p2 -o p2 -- python3 tools/zgemm_one.py 9984 9984 1024 3 > $O/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/p3 -o p3 -- python3 tools/zgemm_one.py 9984 9984 1024 3 > $O/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob("gpurun_out/r03_zgemm_pmc/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "zgemm3m" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc):
            print("%-32s %16.0f per launch (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
