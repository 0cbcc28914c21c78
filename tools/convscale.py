"""How does the 8->8 gather kernel scale with rows (1 round of waves vs several)?  python tools/convscale.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import ops, synthetic, engine
from linr_pcgc_amd.module_utils import prepare_frame
from tools.convlab import timeit
dev = 'cuda'
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev, with_arena=False)
R, ld = f.rows, f.nbr_ld
w = torch.randn(27, 8, 8, device=dev) * 0.1
b = torch.zeros(1, 8, device=dev)
for mult in (0.125, 0.25, 0.5, 1, 2, 4, 8):
    if mult <= 1:
        n = int(R * mult)
        nbr = f.nbr[:, :n].contiguous()
        nbr = torch.where(nbr >= n, torch.full_like(nbr, -1), nbr)
    else:
        m = int(mult)
        n = R * m
        nbr = torch.cat([torch.where(f.nbr[:, :R] >= 0, f.nbr[:, :R] + i * R, f.nbr[:, :R]) for i in range(m)], dim=1).contiguous()
    x = torch.zeros((n + 1, 8), device=dev); x[1:].normal_()
    out = torch.empty((n, 8), device=dev)
    t = timeit(lambda: ops.spconv_fwd(x[1:], nbr, w, b, out=out, pad_row=True))
    print('rows %8d  %.1f us   %.3f ns/row   alg %.0f GB/s' % (n, t, t * 1e3 / n, n * 172 / t / 1e3))
