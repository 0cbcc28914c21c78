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
