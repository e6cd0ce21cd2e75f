#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r04_c5.log; : > $L
run() { env "$@" timeout -k 10 400 python bench.py --steps 6 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config5',{}); print('$*: slfmm %.4f ms  mlfmm %.4f ms  extras_error %s' % (c.get('slfmm',{}).get('apply_ms',-1), c.get('mlfmm',{}).get('apply_ms',-1), d.get('extras_error')))" | tee -a $L; }
run MA_FMM_OVERLAP=1
run MA_FMM_OVERLAP=1 MA_FMM_NEAR_PRIO=-1
run MA_FMM_OVERLAP=0
