#!/bin/bash
# same-box A/B of an environment switch on the default bench:  gpurun -- 'bash tools/ab_env.sh LINR_FUSED_BWD=0 [steps]'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SW=$1; ST=${2:-96}
b(){ LINR_SKIP_ROOFLINE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sequence --steps $ST 2>/tmp/ab_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['ms_per_step'], d['bits_per_point'])
except Exception as e:
    print('FAILED', repr(e)); print(open('/tmp/ab_err.txt').read()[-1500:])"; }
for rep in 1 2 3; do
  echo "default : $(b)"
  echo "$SW : $(env $SW bash -c "$(declare -f b); ST=$ST; b")"
done
