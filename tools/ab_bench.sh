#!/bin/bash
# Same-box A/B of two versions of one csrc file:  gpurun -- 'bash tools/ab_bench.sh fused.hip tools/_a.txt tools/_b.txt'
# (box-to-box spread is ~5 %, so small effects only show when both arms run in one call; arms alternate A B A B)
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; A=$2; B=$3
b(){ (cd $R && LINR_SKIP_ROOFLINE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sequence --steps 96 2>/tmp/ab_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    print(json.loads(t.strip().splitlines()[-1])['ms_per_step'])
except Exception as e:
    print('FAILED', repr(e)); print(open('/tmp/ab_err.txt').read()[-800:])"); }
for arm in A B A B; do
  src=$A; [ $arm = B ] && src=$B
  cp $R/$src $R/linr_pcgc_amd/csrc/$F
  (cd $R && bash linr_pcgc_amd/csrc/build.sh > /dev/null 2>&1)
  echo "$arm ($src): $(b) $(b)"
done
