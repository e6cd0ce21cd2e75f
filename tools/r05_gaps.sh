#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
MA_LU_CU_SPLIT=32 timeout -k 10 300 python bench.py --steps 48 --warmup 3 --no-cpu-baseline --no-extras --no-check --dump-updates gpurun_out/r05_updates.npy > gpurun_out/r05_gaps.json 2> gpurun_out/r05_gaps.err
grep "big updates" gpurun_out/r05_gaps.err
python - <<'PY'
import numpy as np
iv = np.load("gpurun_out/r05_updates.npy")
dur = iv[:,1]-iv[:,0]; gaps = iv[1:,0]-iv[:-1,1]
# steady state: skip first and last 15%
n=len(iv); a=int(0.2*n); b=int(0.8*n)
print("updates", n, "steady window", a, b, "span ms", iv[b,1]-iv[a,0], "busy", dur[a:b].sum(), "gaps", gaps[a:b].sum())
g=gaps[a:b]
for thr in (0.02,0.05,0.1,0.2,0.5,1.0):
    print("  gaps > %.2f ms: %d, total %.1f ms" % (thr,(g>thr).sum(), g[g>thr].sum()))
# gap by position within round (3 updates per round): which of the round's updates waits
k = np.arange(a,b)
for r in range(3):
    sel = g[(k % 3) == r]
    print("  position %d in its round: mean gap %.3f ms, mean dur of the update after it %.3f" % (r, sel.mean(), dur[a+1:b+1][(k%3)==r].mean()))
PY
