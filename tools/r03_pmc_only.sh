#!/bin/bash
# the two PMC passes of tools/profile_r03.sh alone (pipeline schedule: the kernels the default bench runs)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03prof_pmc; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 6 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "$CMD"
rm -rf $O/pmc_fetch $O/pmc_write
