#!/bin/bash
# the XCD-aware tile order of the LDS-DMA update kernel: correctness, kernel alone, sweep, fabric traffic
set -o pipefail
cd "$(dirname "$0")/.."
export PYTHONPATH=. TMPDIR=/tmp
L=gpurun_out/r03_tile_order.log; : > $L
timeout -k 10 300 python -m pytest tests/test_lu_gpu.py -q -x 2>&1 | tail -2 | tee -a $L
for o in 0 1; do
  echo "== MA_ZGEMM_TILE_ORDER=$o: kernel alone" | tee -a $L
  MA_ZGEMM_TILE_ORDER=$o timeout -k 10 200 python tools/zgemm_k_probe.py 2>&1 | grep -v amdgpu.ids | sed -n 3,6p | tee -a $L
  echo "== MA_ZGEMM_TILE_ORDER=$o: sweep" | tee -a $L
  MA_ZGEMM_TILE_ORDER=$o timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
done
O=gpurun_out/r03_tile_pmc; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 6 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "$CMD"
rm -rf $O/pmc_fetch $O/pmc_write
python -c "
import json
p=json.load(open('$O/pmc_traffic.json'))
for k,v in p['kernels'].items():
    if 'zgemm' in k: print(k, v)" | tee -a $L
