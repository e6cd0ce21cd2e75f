#!/bin/bash
# round 5: what the matrix-core near kernel waits for -- SQ counters of slfmm_near_mfma_kernel on the 50k box (one pass per counter set)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=. MA_FMM_OVERLAP=0
O=gpurun_out/r05_fmm5; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 tools/bench_slfmm_box.py > /dev/null 2> $O/p$i.err
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "near_mfma" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("%-32s per launch %.4g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $O/p$i
done
