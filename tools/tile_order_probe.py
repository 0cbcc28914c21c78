"""Does a (x-tile, y-tile)-major internal row order speed the executor up?  (keeps the dz-consecutive property the
compressed kernel map needs).  python tools/tile_order_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import _lib, engine, synthetic, overfit
from linr_pcgc_amd.module_utils import prepare_frame
from tools.convlab import timeit

dev = 'cuda'
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
model = overfit.gen_model(fr['scale_num'], dev, seed=1)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev)
R, ld = f.rows, f.nbr_ld
bits = torch.zeros(1, dtype=torch.float64, device=dev)
flat = model.flat_parameters()
m = torch.zeros_like(flat); v = torch.zeros_like(flat)


def run_fwd():
    engine.net_forward(f, flat, 0, 8, None, bits)


def run_step():
    engine.net_train_step(f, flat, m, v, 1e-6, 1, 0.0, 0.9, 0.999, 1e-8, 0.0, bits)


print('x-major      fwd %.0f us   step %.0f us' % (timeit(run_fwd, 10), timeit(run_step, 10)))
for T in (2, 3, 4, 5):
    perm = torch.empty(R, dtype=torch.int64, device=dev)
    for i, s in enumerate(fr['all_input_info']):
        sl = f.scale_slice(i)
        c = s['coord'].to(torch.int64)
        key = ((c[:, 0] >> T) << 50) | ((c[:, 1] >> T) << 40) | ((c[:, 0] & ((1 << T) - 1)) << 34) | ((c[:, 1] & ((1 << T) - 1)) << 28) | c[:, 2]
        perm[sl] = torch.argsort(key, stable=True) + sl.start
    inv = torch.empty_like(perm); inv[perm] = torch.arange(R, device=dev)
    nb = f_nbr0[:, :R] if 'f_nbr0' in globals() else f.nbr[:, :R].clone()
    if 'f_nbr0' not in globals():
        f_nbr0 = f.nbr.clone(); occ0 = f.occ.clone(); off0 = f.offset_feat.clone()
        nb = f_nbr0[:, :R]
    nb2 = nb[:, perm]
    nb2 = torch.where(nb2 >= 0, inv[nb2.clamp(min=0)].to(torch.int32), nb2)
    f.nbr[:, :R] = nb2
    f.occ.copy_(occ0[perm]); f.offset_feat.copy_(off0[perm])
    _lib.check(_lib.lib().linr_kmap_compress(f.nbr.data_ptr(), ld, R, f.nbr_lo.data_ptr(), f.nbr_mask.data_ptr(), ld,
                                             torch.cuda.current_stream().cuda_stream), 'compress')
    # sanity: dz-consecutive property must hold for the compressed map to be valid
    lo = f.nbr_lo[:, :R].long(); mk = f.nbr_mask[:R].long()
    ok = True
    for q in range(9):
        b0 = (mk >> (3 * q)) & 1; b1 = (mk >> (3 * q + 1)) & 1; b2 = (mk >> (3 * q + 2)) & 1
        ok &= bool(((b0 == 0) | (f.nbr[q, :R].long() == lo[q])).all())
        ok &= bool(((b1 == 0) | (f.nbr[q + 9, :R].long() == lo[q] + b0)).all())
        ok &= bool(((b2 == 0) | (f.nbr[q + 18, :R].long() == lo[q] + b0 + b1)).all())
    print('tile 2^%d cols  fwd %.0f us   step %.0f us   (compressed map valid: %s)' % (T, timeit(run_fwd, 10), timeit(run_step, 10), ok))
