#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for mode in tournament partial; do
  O=gpurun_out/r05_prof2_$mode; rm -rf $O; mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 tools/r05_panel_idle.py $mode > $O/out.txt 2> $O/err.log || exit 1
  cat $O/out.txt
  grep -E "panel|finish|lane_step" $O/t_kernel_stats.csv | cut -c1-60,150-260
  find $O -name "*kernel_trace.csv" -delete
done
