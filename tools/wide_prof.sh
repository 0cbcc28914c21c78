#!/bin/bash
# kernel-time table of the channel-blocked executor (hidden_channel_conv 16):  gpurun -- 'bash tools/wide_prof.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wide
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wide -- python3 $R/tools/wide_speed.py loot10 > /tmp/wide.log 2>&1
grep hidden /tmp/wide.log
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_wide/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms over %d kernel names' % (tot / 1e6, len(rows)))
for r in rows[:14]:
    print('%-70s calls %6s avg %8.1f us total %8.1f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
