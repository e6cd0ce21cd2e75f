#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/r05_prof3 && mkdir -p gpurun_out/r05_prof3
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_prof3 -o t -- python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-extras --no-check --no-timing > gpurun_out/r05_prof3/bench.json 2> gpurun_out/r05_prof3/err.log
python3 -c "import json; d=json.load(open('gpurun_out/r05_prof3/bench.json')); print('ms_per_step under the profiler', d['ms_per_step'])"
f=$(find gpurun_out/r05_prof3 -name "*kernel_trace.csv" | head -1)
head -2 "$f" | cut -c1-400
python3 tools/r05_trace_analysis.py "$f" | tee gpurun_out/r05_prof3/analysis.txt
find gpurun_out/r05_prof3 -name "*kernel_trace.csv" -delete
