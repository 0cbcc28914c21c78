#!/bin/bash
# same-box A/B of lab builds of the library: head probe + the per-class kernel timing of both training executors
#   gpurun -- 'bash tools/ab_classes.sh "<class name filter regex>" tools/_lab/liblinr_x.so ...'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
filt=$1; shift
b(){ python3 tools/bf16_train_speed.py --classes 2>/dev/null | python3 -c "
import json,sys,re
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
for k in ('f32','bf16'):
    print('   %s %.4f ms/step | ' % (k, d[k]['ms_per_step']) + '  '.join('%s %.1f' % (n, c['us_per_step']) for n, c in d[k]['classes'].items() if re.search(sys.argv[1], n)))" "$filt"; }
echo "default : $(b)"
for lib in "$@"; do echo "$lib : $(LINR_HIP_LIB=$R/$lib bash -c "$(declare -f b); filt='$filt'; b")"; done
echo "default : $(b)"
