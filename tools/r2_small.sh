#!/bin/bash
# same-box comparison of the small-frame step (BASELINE config[0] geometry) with / without the second stream, plus loot10
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
b(){ LINR_SKIP_ROOFLINE=1 python bench.py --no-cpu-baseline --no-sequence "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['per_step_ms_hip_events']['median'], d['bits_per_point'])"; }
for rep in 1 2; do
  echo "sphere8 stream=auto : $(b --config sphere8 --gop 8 --steps 80)"
  echo "sphere8 stream=0    : $(LINR_WGRAD_STREAM=0 b --config sphere8 --gop 8 --steps 80)"
done
echo "loot10 default      : $(b --steps 64)"
echo "loot10 stream=1     : $(LINR_WGRAD_STREAM=1 b --steps 64)"
echo "andrew10 default    : $(b --config andrew10 --gop 8 --steps 80)"
