"""Launches linr_spconv_bwd_fused (and the two kernels it replaces) on frame 0 of loot10: hipEvent timing, or a target for
rocprofv3 --pmc (tools/pmc_fused.sh).  usage: fused_probe.py [reps] [nblocks]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import ops, synthetic, engine
from linr_pcgc_amd.module_utils import prepare_frame
dev = 'cuda'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev, with_arena=False)
R = f.rows
x = torch.randn(R, 8, device=dev)
go = torch.zeros((R + 1, 8), device=dev); go[1:].normal_()
w = torch.randn(27, 8, 8, device=dev) * 0.1
lo, mask = f.nbr_lo, f.nbr_mask
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('rows', R)
print('fused bwd+wgrad  %.1f us' % t(lambda: ops.spconv_bwd_fused(go[1:], x, lo, mask, R, w, nblocks=nb, reduce=False)))
print('bwd-data alone   %.1f us' % t(lambda: ops.spconv_cmap(go[1:], lo, mask, R, w, None, bwd=True)))
# both conv pairs of an Inception layer's backward (conv_bwd_wgrad_k<1> then <2>): time them apart with rocprofv3 --kernel-trace --stats
import ctypes
from linr_pcgc_amd import _lib
L = _lib.lib()
gen = torch.Generator().manual_seed(1)
shapes = {'w00': (27, 8, 4), 'b00': (4,), 'w01': (27, 4, 4), 'b01': (4,), 'w10': (8, 4), 'b10': (4,), 'w11': (27, 4, 4), 'b11': (4,), 'w12': (4, 4), 'b12': (4,)}
wd = {k: (torch.randn(*s, generator=gen) * 0.2).to(dev).contiguous() for k, s in shapes.items()}
q = _lib.LinrInceptionParams(**{k: v.data_ptr() for k, v in wd.items()})
def padded(c):
    t = torch.zeros((R + 1, c), device=dev); t[1:].normal_(); return t
gI, gM, xp, H, gH, gX = padded(8), padded(4), padded(8), padded(8), padded(8), padded(8)
slab = torch.zeros((nb, 1776), device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def inc():
    _lib.check(L.linr_inception_bwd_fused(gI[1:].data_ptr(), gM[1:].data_ptr(), xp[1:].data_ptr(), H[1:].data_ptr(), lo.data_ptr(), mask.data_ptr(),
                                          lo.stride(0), R, ctypes.byref(q), gH[1:].data_ptr(), gX[1:].data_ptr(), 2, slab.data_ptr(), nb, st), 'inc')
print('inception bwd (two 4->4 convs, then conv0_0 8->4 + conv1_0)  %.1f us' % t(inc))
