"""Per-row cost of the executor's MFMA conv kernel vs rows per launch (is batching independent layers worth it?)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import ops, synthetic, engine
from linr_pcgc_amd.module_utils import prepare_frame
from tools.convlab import timeit
dev = 'cuda'
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev, with_arena=False)
R = f.rows
w = torch.randn(27, 8, 8, device=dev) * 0.1
b = torch.zeros(8, device=dev)
for m in (1, 2, 4, 8):
    n = R * m
    nbr = torch.cat([torch.where(f.nbr[:, :R] >= 0, f.nbr[:, :R] + i * R, f.nbr[:, :R]) for i in range(m)], dim=1).contiguous()
    lo, mask = ops.kmap_compress(nbr)
    x = torch.zeros((n + 1, 8), device=dev); x[1:].normal_()
    out = torch.empty((n, 8), device=dev)
    t = timeit(lambda: ops.spconv_cmap(x[1:], lo, mask, n, w, b, out=out))
    print('rows %8d  %.1f us   %.4f ns/row' % (n, t, t * 1e3 / n))
print('--- all neighbours absent (pad row only): compute-side bound')
for m in (1, 4):
    n = R * m
    lo = torch.zeros((9, n), dtype=torch.int32, device=dev); mask = torch.zeros((n,), dtype=torch.int32, device=dev)
    x = torch.zeros((n + 1, 8), device=dev); x[1:].normal_()
    out = torch.empty((n, 8), device=dev)
    t = timeit(lambda: ops.spconv_cmap(x[1:], lo, mask, n, w, b, out=out))
    print('rows %8d  %.1f us   %.4f ns/row' % (n, t, t * 1e3 / n))
