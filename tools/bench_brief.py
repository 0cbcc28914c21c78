"""One screen of a bench.py line: python tools/bench_brief.py <file with the JSON line>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step %.4f  value %.5f s/frame  bpp %.5f  coded epoch %s' % (d['ms_per_step'], d['value'], d['bits_per_point'], d.get('coded_epoch')))
print('components', d.get('components_s_per_frame'))
r = d.get('roofline') or {}
if r:
    print('dominant frac %.4f (%.1f us/launch), step frac %.4f' % (r['frac'], r['avg_launch_us'], r['step']['frac']))
    for k in r.get('kernels', []):
        print('  %-62s %7.1f us/step' % (k['kernel'][:62], k['us_per_step']))
if d.get('sequence'):
    print('sequence', {k: d['sequence'].get(k) for k in ('sec_per_frame', 'phase_b_efficiency', 'bits_per_point', 'lossless', 'error')})
