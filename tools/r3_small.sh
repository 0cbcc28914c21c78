#!/bin/bash
# small / dense frames with and without the fused backward:  gpurun -- 'bash tools/r3_small.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
b(){ LINR_SKIP_ROOFLINE=1 LINR_SKIP_BPP_SEEDS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sequence --config $1 --gop 8 --steps 80 2>/tmp/ab_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['ms_per_step'], d['bits_per_point'], d['lossless_decode_frames0to3'])
except Exception as e:
    print('FAILED', repr(e)); print(open('/tmp/ab_err.txt').read()[-800:])"; }
for cfg in sphere8 andrew10; do
  echo "$cfg fused   : $(b $cfg)"
  echo "$cfg unfused : $(LINR_FUSED_BWD=0 bash -c "$(declare -f b); b $cfg")"
done
