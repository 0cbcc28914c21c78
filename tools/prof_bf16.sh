#!/bin/bash
# rocprofv3 kernel statistics (and optional counters) of the bf16 training step:  gpurun -- 'bash tools/prof_bf16.sh [tag] [steps]'
# -> gpurun_out/prof_<tag>/kernel_stats.csv + a per-step table on stdout
TAG=${1:-bf16}; STEPS=${2:-96}; PREC=${3:-bf16}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/tools/bf16_steps.py $PREC $STEPS > /tmp/b_$TAG.log 2>&1
tail -1 /tmp/b_$TAG.log | cut -c1-200
mkdir -p $R/gpurun_out/prof_$TAG
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_$TAG/kernel_stats.csv \;
python3 - $R/gpurun_out/prof_$TAG/kernel_stats.csv $STEPS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = 0.0
print('%-90s %8s %10s %10s' % ('kernel', 'calls/st', 'avg us', 'us/step'))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    us = float(r['TotalDurationNs']) / 1e3 / steps
    if us < 0.5:
        continue
    tot += us
    print('%-90s %8.2f %10.2f %10.1f' % (r['Name'][:90], float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, us))
print('sum of the listed kernels: %.1f us per step' % tot)
PY
