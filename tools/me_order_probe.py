"""Which kernel-offset ordering did MinkowskiEngine use?  Load the REFERENCE-TRAINED weights (loot/gop_32_62/model.pth, kept
as the flat vector of tests/golden/loot_model_kat.npz) and code a synthetic surface with the taps interpreted in the
assumed order k = (dx+1) + 3(dy+1) + 9(dz+1) (x fastest) and in the alternatives.  A network trained on real data only
predicts well when its 27 taps are applied to the neighbours it was trained with."""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit, synthetic          # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'loot_model_kat.npz'), allow_pickle=True)
flat = torch.from_numpy(g['flat'].astype(np.float32))
for cfg in (sys.argv[1:] or ['loot10', 'andrew10']):          # e.g. `me_order_probe.py loot10_rough`: the non-spherical figure
    pts = synthetic.sequence_frame_device(cfg, 0, 'cuda')
    gop = overfit.Gop(None, [pts], 7, 64, 'cuda')
    model = overfit.gen_model(7, 'cuda', seed=8807)
    _, b0 = model.frame_probs(gop.frames[0])
    print('%s: untrained (seed 8807)                 %.4f bpp' % (cfg, float(b0) / gop.point_nums[0]))
    sd = model.state_dict()
    names = list(sd.keys())
    assert names == [str(n) for n in g['names']]

    def load(perm_axes, flip):
        """tap k = (d[a0]+1) + 3 (d[a1]+1) + 9 (d[a2]+1) in the candidate order; flip: correlation vs convolution"""
        off = 0
        new = {}
        for n in names:
            t = sd[n]
            w = flat[off:off + t.numel()].view(t.shape).clone()
            off += t.numel()
            if w.dim() == 3 and w.shape[0] == 27:
                idx = []
                for k in range(27):                       # our tap k = (dx+1)+3(dy+1)+9(dz+1)
                    d = [k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1]
                    if flip:
                        d = [-v for v in d]
                    idx.append((d[perm_axes[0]] + 1) + 3 * (d[perm_axes[1]] + 1) + 9 * (d[perm_axes[2]] + 1))
                w = w[idx]
            new[n] = w
        model.load_state_dict(new)

    for perm in itertools.permutations(range(3)):
        for flip in (False, True):
            load(perm, flip)
            _, b = model.frame_probs(gop.frames[0])
            tag = 'axis order %s fastest-first%s' % (''.join('xyz'[a] for a in perm), ', mirrored' if flip else '')
            print('%s: reference weights, %-40s %.4f bpp' % (cfg, tag, float(b) / gop.point_nums[0]))
