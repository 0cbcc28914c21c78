#!/bin/bash
# Kernel-time table of the training step on the GPU box:  gpurun -- 'bash tools/prof_step.sh [tag]'
# Writes gpurun_out/prof_<tag>/kernel_stats.csv and prints the per-step table of the library's own kernels.
TAG=${1:-step}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-sequence --steps 64 --warmup 32 --ramp-s 0 > /tmp/b_$TAG.log 2>&1
tail -1 /tmp/b_$TAG.log | cut -c1-180
mkdir -p $R/gpurun_out/prof_$TAG
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_$TAG/kernel_stats.csv \;
find /tmp/prof_$TAG -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/prof_$TAG/kernel_trace.csv \;
python3 $R/tools/step_table.py $R/gpurun_out/prof_$TAG/kernel_stats.csv 96
