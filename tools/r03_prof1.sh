#!/bin/bash
# round 3: kernel traces of (1) one 10k factorisation with every launch on one stream (idle durations of each kernel), old and register
# panel kernel; (2) the sweep with the register panel kernel, without and with the CU split
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03prof1; rm -rf $O; mkdir -p $O
for mode in 0 1; do
  MA_LU_LOOKAHEAD=0 MA_LU_REG_PANEL=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/single_reg$mode -- python3 tools/lu_big_random.py 10000 > $O/single_reg$mode.json 2> $O/single_reg$mode.err || exit 1
  f=$(find $O/single_reg$mode -name "*kernel_stats.csv" | head -1); cp $f $O/single_reg${mode}_kernel_stats.csv
done
for cfg in "reg1_split0 MA_LU_REG_PANEL=1" "reg1_split32 MA_LU_REG_PANEL=1 MA_LU_CU_SPLIT=32"; do
  set -- $cfg; name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-timing > $O/$name.json 2> $O/$name.err || exit 1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); cp $f $O/${name}_kernel_stats.csv
  t=$(find $O/$name -name "*kernel_trace.csv" | head -1)
  python tools/chain_analysis.py $t $O/${name}_chain.json > /dev/null 2>&1
  python tools/kernel_timeline.py $t 3000 > $O/${name}_timeline.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; head -12 $O/single_reg0_kernel_stats.csv; head -12 $O/single_reg1_kernel_stats.csv; head -14 $O/reg1_split32_kernel_stats.csv
