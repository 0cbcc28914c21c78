"""Per-step table of the library's kernels from a rocprofv3 *_kernel_stats.csv:  step_table.py <csv> <steps in the run>."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
mine = [r for r in rows if not r['Name'].startswith('void at::') and 'rocclr' not in r['Name']]
tot = 0.0
for r in mine:
    per = float(r['TotalDurationNs']) / steps / 1e3
    tot += per
    print('%-78s calls/step %6.2f avg %8.1f us  per-step %8.1f us' % (r['Name'][:78], int(r['Calls']) / steps,
                                                                       float(r['AverageNs']) / 1e3, per))
print('sum of own kernels per step (incl. the encode / decode legs): %.1f us' % tot)
