#!/bin/bash
# Floors of the 8->8 weight-gradient kernel inside the real training step (grouped launches), from lab builds of fused.hip
# (tools/wgrad_floor_lab.py writes + builds them here; WG_LAB: 0 full, 1 no gathers, 2 gathers + 1 of 8 MFMAs, 3 no index /
# gradient loads (arithmetic indices), 4 MFMAs + loop only, 5 no cin_valid branches).  gpurun -- 'bash tools/wgrad_floor_lab.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2 3 4 5}; do
  export LINR_HIP_LIB=$R/tools/_lab/liblinr_lab$v.so
  rm -rf /tmp/lab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lab_$v -- python3 $R/bench.py --no-cpu-baseline --no-sequence --gop 4 --steps 40 --warmup 8 --ramp-s 0 > /tmp/lab_$v.log 2>&1 || { echo "variant $v failed"; tail -5 /tmp/lab_$v.log; }
  f=$(find /tmp/lab_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" /tmp/lab_$v.log <<'P'
import csv, json, sys
v, f, log = sys.argv[1:4]
ms = None
for line in open(log):
    if line.startswith('{'):
        ms = json.loads(line).get('ms_per_step')
for r in csv.DictReader(open(f)):
    if 'spconv_wgrad_mfma_k<2, 8, false, 3>' in r['Name']:
        print('WG_LAB=%s  wgrad<2,8,3>: calls %s avg %.1f us   step %s ms' % (v, r['Calls'], float(r['AverageNs']) / 1e3, ms))
P
done
