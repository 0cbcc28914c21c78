"""Launches the two kernels bench.py's roofline reports (executor configuration, frame 0 of loot10) a few times so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can attribute HBM traffic to them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import engine, ops, synthetic                    # noqa: E402
from linr_pcgc_amd.module_utils import prepare_frame                # noqa: E402

dev = 'cuda'
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev, with_arena=False)
R = f.rows
x = torch.zeros((R + 1, 8), device=dev); x[1:].normal_()
g = torch.randn((R, 8), device=dev)
out = torch.empty((R, 8), device=dev)
w = torch.randn(27, 8, 8, device=dev) * 0.1
b = torch.zeros(8, device=dev)
slab = torch.empty((512, 27 * 64 + 8), device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.spconv_wgrad_cmap(x[1:], g, f.nbr, None, None, R, 8, 8, slab=slab, reduce=False, tile8t=f.nbr8t)
    ops.spconv_cmap(x[1:], f.nbr_lo, f.nbr_mask, R, w, b, out=out)
torch.cuda.synchronize()
print('rows', R)
