"""Training steps with every launch preceded by the on-chip poison kernel (linr_debug_poison): which parameters differ / are not finite?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit, synthetic, _lib
from linr_pcgc_amd.model_core import FlatAdam, train_step
cfg = sys.argv[1] if len(sys.argv) > 1 else 'loot10'
clouds = [synthetic.sequence_frame_device(cfg, t, 'cuda') for t in range(2)]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
L = _lib.lib()
def run(poison, steps=6):
    m = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    o = FlatAdam(m)
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    L.linr_debug_poison(poison)
    tr = []
    for s in range(steps):
        bits.zero_()
        train_step(m, o, gop.frames[s % 2], gop.point_nums[s % 2], out=bits)
        tr.append(float(bits))
    L.linr_debug_poison(0)
    torch.cuda.synchronize()
    return m, o, tr
m0, o0, t0 = run(0, 2)
sd0 = m0.state_dict()
names = ['fused88', 'conv88 fwd', 'fused dual44', 'fused conv84', 'head fwd', 'convpw fwd', 'dual fwd', 'occ7', 'head bwd', 'wgrad_t', 'lin wgrad', 'sce', 'misc', 'bwd data']
for kind in range(14):
    m1, o1, t1 = run(1 << kind, 2)
    sd1 = m1.state_dict()
    bad = [(k, int((~torch.isfinite(sd1[k])).sum()), float((sd0[k] - sd1[k]).abs().max())) for k in sd0 if not torch.equal(sd0[k], sd1[k])]
    print('kind %2d %-14s bits %s: %d of %d tensors differ%s' % (kind, names[kind], 'same' if t0 == t1 else 'DIFFER', len(bad), len(sd0),
          (': ' + ', '.join('%s (nan %d, %.2g)' % b for b in bad[:4])) if bad else ''))
