#!/bin/bash
# kernel-only durations of tools/wgrad_lab.py under rocprofv3 for the LINR_WG_VAR variants
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in ${VARS:-0 1 2}; do
  rm -rf /tmp/wp_$v
  LINR_WG_VAR=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp_$v -- python3 $R/tools/wgrad_lab.py 10 > /tmp/wp_$v.log 2>&1
  echo "== VAR $v"
  python3 $R/tools/step_table.py $(find /tmp/wp_$v -name "*kernel_stats.csv") 1 | grep -E "wgrad|cconv|slab" | cut -c1-60,78-200
done
