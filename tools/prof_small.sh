#!/bin/bash
# Kernel-time table of the small-frame training step (BASELINE config[0] geometry):  gpurun -- 'bash tools/prof_small.sh [tag]'
TAG=${1:-small}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
LINR_SKIP_ROOFLINE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-sequence --config sphere8 --gop 8 --steps 80 --ramp-s 0 > /tmp/b_$TAG.log 2>&1
tail -1 /tmp/b_$TAG.log | cut -c1-180
mkdir -p $R/gpurun_out/prof_$TAG
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_$TAG/kernel_stats.csv \;
find /tmp/prof_$TAG -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/prof_$TAG/kernel_trace.csv \;
python3 $R/tools/step_table.py $R/gpurun_out/prof_$TAG/kernel_stats.csv 80
