#!/bin/bash
# the LDS-DMA update kernel (MA_ZGEMM_DMA=1: 64 x 128 tiles, 4 wavefronts; =2: 128 x 128, 8 wavefronts) against the round-2 kernel
set -o pipefail
cd "$(dirname "$0")/.."
export PYTHONPATH=.
L=gpurun_out/r03_dma_gemm.log; : > $L
for mode in 1 2; do
  echo "== MA_ZGEMM_DMA=$mode: tests/test_lu_gpu.py" | tee -a $L
  MA_ZGEMM_DMA=$mode timeout -k 10 300 python -m pytest tests/test_lu_gpu.py -q -x 2>&1 | tail -3 | tee -a $L || exit 1
done
for mode in 0 1 2; do
  echo "== MA_ZGEMM_DMA=$mode: kernel alone over K" | tee -a $L
  MA_ZGEMM_DMA=$mode timeout -k 10 200 python tools/zgemm_k_probe.py 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $L
done
for mode in 0 1 2; do
  echo "== MA_ZGEMM_DMA=$mode: bench" | tee -a $L
  MA_ZGEMM_DMA=$mode timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-timing --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step %.2f' % d['ms_per_step'])" | tee -a $L
done
