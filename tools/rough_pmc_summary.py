"""Side-by-side counter table of tools/rough_pmc.sh: per kernel of a training step, sphere (loot10) against rough figure (loot10_rough):
HBM bytes per row (2 x FETCH_SIZE + WRITE_SIZE, KB counters), vector-L1 hit rate = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES, texture
addresser busy = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8) (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs), and the kernel's mean duration per row (from the FETCH_SIZE pass's kernel trace)."""
import collections
import csv
import glob
import re
import sys

prec, bases = sys.argv[1], sys.argv[2:4]


def key(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').strip()


def counters(d):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not fs:
        return agg
    for r in csv.DictReader(open(fs[0])):
        k = key(r['Kernel_Name'])
        if re.search(r'_k(<.*>)?$', k):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def durations(d):
    fs = glob.glob(d + '/*/*kernel_trace.csv')
    agg = collections.defaultdict(list)
    if not fs:
        return agg
    for r in csv.DictReader(open(fs[0])):
        agg[key(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return agg


def rows_of(base):
    try:
        for line in open(base + '_1.log'):
            if line.startswith('rows'):
                return int(line.split()[1])
    except OSError:
        pass
    return None


tab = {}
for base in bases:
    c = collections.defaultdict(dict)
    for i in (1, 2, 3, 4):
        for k, d in counters('%s_%d' % (base, i)).items():
            for n, v in d.items():
                c[k][n] = sum(v) / len(v)
                c[k]['n'] = len(v)
    for k, v in durations(base + '_1').items():
        if k in c:
            c[k]['us'] = sum(v) / len(v)
    tab[base] = (rows_of(base), c)

(r0, c0), (r1, c1) = tab[bases[0]], tab[bases[1]]
print('== %s executor: frame 0 of loot10 (%s rows) | loot10_rough (%s rows); per dispatch of the kernel, 3 training steps' % (prec, r0, r1))
print('%-44s %5s | %9s %9s | %7s %7s | %7s %7s | %9s %9s %6s' % ('kernel', 'n', 'B/row sph', 'B/row rgh', 'L1 sph', 'L1 rgh', 'TA sph', 'TA rgh',
                                                                'ps/row sph', 'ps/row rgh', 'ratio'))


def fig(c, rows):
    b = (2048.0 * c.get('FETCH_SIZE', 0.0) + 1024.0 * c.get('WRITE_SIZE', 0.0)) / rows if rows else float('nan')
    tot = c.get('TCP_TOTAL_CACHE_ACCESSES_sum')
    l1 = 1.0 - c.get('TCP_TCC_READ_REQ_sum', 0.0) / tot if tot else float('nan')
    gui = c.get('GRBM_GUI_ACTIVE')
    ta = c.get('TA_BUSY_avr', float('nan')) / (gui / 8.0) if gui else float('nan')
    return b, l1, ta, (1e6 * c['us'] / rows if 'us' in c and rows else float('nan'))


for k in sorted(c0, key=lambda k: -c0[k].get('us', 0.0) * c0[k].get('n', 0)):
    if k not in c1 or c0[k].get('us', 0.0) * c0[k].get('n', 0) < 15.0:
        continue
    a, b = fig(c0[k], r0), fig(c1[k], r1)
    print('%-44s %5d | %9.1f %9.1f | %7.3f %7.3f | %7.3f %7.3f | %9.2f %9.2f %6.3f' % (k[:44], c0[k].get('n', 0), a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3],
                                                                                      b[3] / a[3] if a[3] == a[3] and a[3] else float('nan')))
print('(B/row = HBM bytes of one dispatch / rows of the frame - a grouped launch covers several row passes; ps/row = mean dispatch duration / rows)')
