"""Does a training step stay finite / reproducible while ANOTHER process trains on the same GPU?  Start two copies at once
(tools/concurrency_probe.sh).  Each trains a 4-frame GOP from scratch, then warm-starts fresh models from that state and trains on,
checking after every step that parameters and moments are finite, and that two runs from the same state give the same bits."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit, synthetic
from linr_pcgc_amd.model_core import FlatAdam, train_step
tag = sys.argv[1] if len(sys.argv) > 1 else 'p'
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
clouds = [synthetic.sequence_frame_device('loot10', t, 'cuda') for t in range(4)]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
opt = FlatAdam(model)
overfit.overfit_gop(model, opt, gop, 2)
ck = overfit.checkpoint(model, opt, 1, 0.0)
ref = None
for rnd in range(rounds):
    m = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    o = FlatAdam(m)
    overfit.warm_start(m, o, ck)
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    trace = []
    bad = None
    for step in range(24):
        f = gop.frames[step % 4]
        bits.zero_()
        train_step(m, o, f, gop.point_nums[step % 4], out=bits)
        torch.cuda.synchronize()
        trace.append(float(bits))
        fin = [bool(torch.isfinite(t).all()) for t in (m.flat_parameters(), o.exp_avg, o.exp_avg_sq)]
        if not all(fin) or not np.isfinite(trace[-1]):
            bad = (step, fin, trace[-1])
            break
    if ref is None:
        ref = trace
    same = trace == ref[:len(trace)]
    print('[%s] round %d: %s, bits trace %s the first round; last bits %.1f' %
          (tag, rnd, 'NON-FINITE at step %d (params, m, v finite: %s; bits %r)' % bad if bad else 'finite', 'equals' if same else 'DIFFERS from', trace[-1]), flush=True)
    if not same:
        d = [i for i, (a, b) in enumerate(zip(trace, ref)) if a != b]
        print('[%s]   first differing step %d: %.3f vs %.3f' % (tag, d[0], trace[d[0]], ref[d[0]]), flush=True)
