#!/bin/bash
# kernel trace of the staged pipeline in pair mode under the CU split: per-queue busy time and per-kernel durations under load
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_trace; rm -rf $O; mkdir -p $O
MA_LU_REG_PANEL=${REG:-2} MA_LU_CU_SPLIT=${SPLIT:-64} timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -o trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-timing --no-extras > $O/bench.json 2> $O/err.log
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/trace_queues.py "$f" > $O/summary.txt
rm -f "$f"
cat $O/summary.txt
