"""Micro-experiments on the sparse-conv gather kernel (not part of the product): where does its time go?
Runs on the GPU box:  python tools/convlab.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import ops, synthetic, engine
from linr_pcgc_amd.module_utils import prepare_frame


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def morton_key(c):
    def spread(v):
        v = v.astype(np.uint64)
        out = np.zeros_like(v)
        for b in range(11):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out
    return (spread(c[:, 0]) << np.uint64(2)) | (spread(c[:, 1]) << np.uint64(1)) | spread(c[:, 2])


def main():
    dev = 'cuda'
    pts = synthetic.sequence_frame('loot10', 0)
    fr = prepare_frame(pts, None, 64, device=dev)
    f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev, with_arena=False)
    R, ld = f.rows, f.nbr_ld
    print('rows', R)
    x = torch.zeros((R + 1, 8), device=dev); x[1:].normal_()
    out = torch.empty((R, 8), device=dev)
    w = torch.randn(27, 8, 8, device=dev) * 0.1
    b = torch.zeros(1, 8, device=dev)
    nbr = f.nbr
    base = timeit(lambda: ops.spconv_fwd(x[1:], nbr, w, b, out=out, pad_row=True))
    print('V0 baseline PAD            %.1f us' % base)
    print('V0b branch version         %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], nbr, w, b, out=out, pad_row=False)))
    # no gather: every neighbour = own row
    self_nbr = torch.arange(ld, device=dev, dtype=torch.int32).clamp(max=R - 1).repeat(27, 1).contiguous()
    print('V2 self-neighbours (VALU)  %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], self_nbr, w, b, out=out, pad_row=True)))
    # all absent: reads only the zero row
    none_nbr = torch.full((27, ld), -1, device=dev, dtype=torch.int32)
    print('V3 all-absent (pad row)    %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], none_nbr, w, b, out=out, pad_row=True)))
    print('V3b all-absent branch      %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], none_nbr, w, b, out=out, pad_row=False)))
    # random neighbours: worst locality
    rnd = torch.randint(0, R, (27, ld), device=dev, dtype=torch.int32)
    print('V4 random neighbours       %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], rnd, w, b, out=out, pad_row=True)))
    # copy kernels for scale
    a = torch.empty(R * 8, device=dev); c = torch.empty(R * 8, device=dev)
    print('copy [R,8] f32             %.1f us' % timeit(lambda: c.copy_(a)))
    big = torch.empty(64 << 20, device=dev); big2 = torch.empty(64 << 20, device=dev)
    t = timeit(lambda: big2.copy_(big)); print('copy 256MB                 %.1f us  (%.2f TB/s r+w)' % (t, 2 * big.numel() * 4 / t / 1e6))
    # Morton order within each scale
    nbr_h = nbr[:, :R].t().cpu().numpy()          # [R,27]
    perm = np.empty(R, dtype=np.int64)            # new -> old
    for i, s in enumerate(fr['all_input_info']):
        sl = f.scale_slice(i)
        c3 = s['coord'].cpu().numpy()
        perm[sl] = np.argsort(morton_key(c3), kind='stable') + sl.start
    inv = np.empty(R, dtype=np.int64); inv[perm] = np.arange(R)
    nb2 = nbr_h[perm]
    nb2 = np.where(nb2 >= 0, inv[np.clip(nb2, 0, None)], -1).astype(np.int32)
    nbr_m = torch.full((27, ld), -1, dtype=torch.int32, device=dev)
    nbr_m[:, :R] = torch.from_numpy(nb2.T.copy()).to(dev)
    print('V5 Morton-ordered rows     %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], nbr_m, w, b, out=out, pad_row=True)))
    print('V5b Morton branch          %.1f us' % timeit(lambda: ops.spconv_fwd(x[1:], nbr_m, w, b, out=out, pad_row=False)))
    # wave-level sparsity statistics: how many of the 27 offsets have ANY present lane per 64-row wave
    for name, tab in (('x-major', nbr_h), ('morton', nb2)):
        pres = (tab >= 0)
        nw = R // 64
        anyp = pres[:nw * 64].reshape(nw, 64, 27).any(axis=1).sum(axis=1)
        print('%s: K_eff per row %.2f, offsets with any lane present per wave %.2f' % (name, pres.sum(1).mean(), anyp.mean()))
    # bwd data + wgrad
    go = torch.zeros((R + 1, 8), device=dev); go[1:].normal_()
    print('bwd_data 8x8               %.1f us' % timeit(lambda: ops.spconv_bwd_data(go[1:], nbr, w, out=out, pad_row=True)))
    print('bwd_data 8x8 morton        %.1f us' % timeit(lambda: ops.spconv_bwd_data(go[1:], nbr_m, w, out=out, pad_row=True)))
    print('wgrad 8x8 (incl. reduce)   %.1f us' % timeit(lambda: ops.spconv_bwd_weight(x[1:], go[1:], nbr, 8, 8)))
    print('wgrad 8x8 morton           %.1f us' % timeit(lambda: ops.spconv_bwd_weight(x[1:], go[1:], nbr_m, 8, 8)))


if __name__ == '__main__':
    main()
