#!/bin/bash
# Same-box A/B of two versions of one csrc file:  gpurun -- 'bash tools/ab_bench.sh fused.hip tools/_a.txt tools/_b.txt'
# (box-to-box spread is ~5 %, so small effects only show when both arms run in one call; arms alternate A B A B)
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; A=$2; B=$3
b(){ (cd $R && LINR_SKIP_ROOFLINE=1 python bench.py --no-cpu-baseline --no-sequence 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for arm in A B A B; do
  src=$A; [ $arm = B ] && src=$B
  cp $R/$src $R/linr_pcgc_amd/csrc/$F
  (cd $R && bash linr_pcgc_amd/csrc/build.sh > /dev/null 2>&1)
  echo "$arm ($src): $(b) $(b)"
done
