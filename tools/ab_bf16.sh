#!/bin/bash
# same-box A/B of lab builds of the library on the bf16 training step:  gpurun -- 'bash tools/ab_bf16.sh tools/_lab/liblinr_x.so ...'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
b(){ python tools/bf16_train_speed.py 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):]); print(d['bf16']['ms_per_step'], d['f32']['ms_per_step'], d['bf16']['bpp_frame0_now'])"; }
echo "default : $(b)"
for lib in "$@"; do echo "$lib : $(LINR_HIP_LIB=$R/$lib bash -c "$(declare -f b); b")"; done
echo "default : $(b)"
