#!/bin/bash
# same-box A/B of the fused backward kernels: single-stream (LINR_FUSED_SPLIT=0) vs wave-specialised (default):  gpurun -- 'bash tools/ab_split.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
b(){ LINR_SKIP_ROOFLINE=1 LINR_SKIP_BPP_SEEDS=1 LINR_SKIP_BF16_TRAIN=1 LINR_SKIP_WIDE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sequence --steps 96 2>/tmp/ab_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['ms_per_step'], d['bits_per_point'])
except Exception as e:
    print('FAILED', repr(e)); print(open('/tmp/ab_err.txt').read()[-800:])"; }
echo "split  : $(b)"
echo "single : $(LINR_FUSED_SPLIT=0 bash -c "$(declare -f b); b")"
echo "split  : $(b)"
echo "single : $(LINR_FUSED_SPLIT=0 bash -c "$(declare -f b); b")"
