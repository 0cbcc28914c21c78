#!/bin/bash
# same-box comparison of several environment settings on one bench configuration:
#   gpurun -- 'bash tools/ab_multi.sh "<bench args>" "" "LINR_A=1" "LINR_A=1 LINR_B=0" ...'     ("" = default; 2 rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ARGS=$1; shift
b(){ LINR_SKIP_ROOFLINE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sequence $ARGS 2>/tmp/ab_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['ms_per_step'], d['per_step_ms_hip_events']['median'], d['bits_per_point'])
except Exception as e:
    print('FAILED', repr(e)); print(open('/tmp/ab_err.txt').read()[-1500:])"; }
for rep in 1 2; do
  for sw in "$@"; do
    echo "[$ARGS] ${sw:-default} : $(env $sw bash -c "$(declare -f b); ARGS='$ARGS'; b")"
  done
done
