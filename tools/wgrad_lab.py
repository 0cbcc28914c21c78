"""Times the backward-weight kernels on frame 0 of loot10 (336,529 rows): op-level entry with pad rows = the executor's
matrix-core kernel + its slab reduce.  usage: python tools/wgrad_lab.py [reps]"""
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
go = torch.zeros((R + 1, 8), device=dev); go[1:].normal_()
w = torch.randn(27, 8, 8, device=dev) * 0.1
b = torch.zeros(8, device=dev)
out = torch.empty((R, 8), device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print('rows', R)
print('wgrad 8x8 mfma (+reduce) %.1f us' % timeit(lambda: ops.spconv_bwd_weight(x[1:], go[1:], f.nbr[:, :R], 8, 8, pad_row=True)))
print('wgrad 8x4 mfma (+reduce) %.1f us' % timeit(lambda: ops.spconv_bwd_weight(x[1:], go[1:], f.nbr[:, :R], 8, 4, pad_row=True)))
print('wgrad 8x8 valu (+reduce) %.1f us' % timeit(lambda: ops.spconv_bwd_weight(x[1:], go[1:], f.nbr[:, :R], 8, 8)))
print('conv fwd 8x8 cmap        %.1f us' % timeit(lambda: ops.spconv_cmap(x[1:], f.nbr_lo, f.nbr_mask, R, w, b, out=out)))
# the executor's variant: compressed map indices (through one training step's kernels is the only C-ABI route, so time
# the whole backward here instead)
