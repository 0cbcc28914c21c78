#!/bin/bash
# conv kernel floors: LINR_CONV_DIAG=1 (MFMAs only, no gathers), 2 (gathers + 1 MFMA per offset)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  rm -rf /tmp/cd_$v
  LINR_CONV_DIAG=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cd_$v -- python3 $R/tools/wgrad_lab.py 10 > /tmp/cd_$v.log 2>&1
  echo "== DIAG $v"
  python3 $R/tools/step_table.py $(find /tmp/cd_$v -name "*kernel_stats.csv") 1 | grep -E "cconv" | cut -c1-60,78-200
done
