#!/bin/bash
# round 5: kernel trace of the tournament sweep (per-kernel durations)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/r05_prof1 && mkdir -p gpurun_out/r05_prof1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_prof1 -o t -- python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-extras --no-check > gpurun_out/r05_prof1/bench.json 2> gpurun_out/r05_prof1/err.log
find gpurun_out/r05_prof1 -name "*kernel_stats*" | head
f=$(find gpurun_out/r05_prof1 -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-220
find gpurun_out/r05_prof1 -name "*kernel_trace.csv" -delete
find gpurun_out/r05_prof1 -name "*.db" -delete
