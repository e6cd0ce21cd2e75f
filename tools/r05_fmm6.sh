#!/bin/bash
# diagnostic: the matrix-core near kernel with only one of its two products (timing only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=. MA_FMM_OVERLAP=0
O=gpurun_out/r05_fmm6; rm -rf $O; mkdir -p $O
for dm in 0 1 2; do
  export MA_TMP_MFMA_PART=$dm
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$dm -- python3 tools/bench_slfmm_box.py > /dev/null 2> $O/p$dm.err
  f=$(find $O/p$dm -name "*kernel_stats.csv" | head -1)
  echo "part $dm: $(grep near_mfma $f | awk -F'",' '{print $2}' | cut -d, -f1-3)"
  rm -rf $O/p$dm
done
