"""Per-kernel HBM bytes per dispatch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in KB); writes traffic.json."""
import collections, csv, glob, json, re, sys

# once-per-frame staging kernels (kernel map, octree) that a profiled run also contains: not part of a training step
STAGING = ('kmap_', 'su_', 'minmax_k', 'octree_', 'occ_bf16_k')


def kernel_key(name):
    """'void (anonymous namespace)::xtg_wgrad_k<2, 1>(Args...)' -> 'void xtg_wgrad_k<2, 1>': the anonymous-namespace marker holds the
    first '(' of such a name, so cutting at it collapsed those kernels into the single key 'void ' (ADVICE r4)."""
    return name.replace('(anonymous namespace)::', '').split('(')[0].strip()


def load(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            key = kernel_key(r['Kernel_Name'])
            bare = key[5:] if key.startswith('void ') else key
            # the library's own kernels are all named *_k (torch's *_kernel and rocprim's match a plain "_k" too)
            if not re.search(r'_k(<.*>)?$', bare) or bare.startswith(STAGING):
                continue
            agg[key].append(float(r['Counter_Value']))
    return agg


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = int(sys.argv[5].split()[1]) if len(sys.argv) > 5 and sys.argv[5].split()[:1] == ['rows'] else None
out = {'steps': steps, 'rows': rows, 'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/traffic_probe.py: 3 training steps of the '
                 'executor on frame 0 of loot10, every dispatch of a step; bytes = 2 x FETCH_SIZE (gfx950: the counter tallies '
                 '128-B requests as 64 B) + WRITE_SIZE, mean per dispatch of that kernel name', 'kernels': {}}
for name in sorted(fetch):
    f, w = fetch[name], write.get(name, [0.0])
    fb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)
    out['kernels'][name] = {'dispatches': len(f), 'dispatches_per_step': round(len(f) / steps, 3), 'bytes_per_step': int((fb + wb) * len(f) / steps), 'fetch_bytes_x2': int(fb), 'write_bytes': int(wb), 'bytes_per_dispatch': int(fb + wb)}
    print('%-60s n=%3d  2xFETCH %8.1f MB  WRITE %7.1f MB' % (name[:60], len(f), fb / 1e6, wb / 1e6))
out['bytes_per_step_all_kernels'] = int(sum(k['bytes_per_step'] for k in out['kernels'].values()))
# which build the counters describe: bench.py repeats this beside the stored bytes (roofline.traffic_note)
import os, time
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'linr_pcgc_amd', 'liblinr_hip.so')
if os.path.exists(lib):
    out['library'] = {'file': 'linr_pcgc_amd/liblinr_hip.so', 'bytes': os.path.getsize(lib),
                      'built_utc': time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime(os.path.getmtime(lib)))}
out['collected_utc'] = time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime())
out['precision'] = sys.argv[6] if len(sys.argv) > 6 else 'f32'
print('all kernels: %.3f GB per training step' % (out['bytes_per_step_all_kernels'] / 1e9))
json.dump(out, open(sys.argv[3], 'w'), indent=1)
