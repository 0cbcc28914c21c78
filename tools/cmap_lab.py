"""Where does the executor's MFMA conv kernel spend its time?  Same kernel, doctored kernel maps."""
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
x = torch.zeros((R + 1, 8), device=dev); x[1:].normal_()
out = torch.empty((R, 8), device=dev)
w = torch.randn(27, 8, 8, device=dev) * 0.1
b = torch.zeros(8, device=dev)
w4 = torch.randn(27, 4, 4, device=dev) * 0.1; b4 = torch.zeros(4, device=dev); out4 = torch.empty((R, 4), device=dev)
lo, mask = f.nbr_lo, f.nbr_mask
zero_mask = torch.zeros_like(mask)
centre = torch.full_like(mask, 1 << (3 * 4 + 1))           # only k = 13 (self) present
lo_self = torch.arange(ld, device=dev, dtype=torch.int32).clamp(max=R - 1).repeat(9, 1).contiguous()
full = torch.full_like(mask, (1 << 27) - 1)                 # every offset "present": lo.. lo+2 consecutive rows
lo_near = (torch.arange(ld, device=dev, dtype=torch.int32).clamp(max=R - 4)).repeat(9, 1).contiguous()
for name, l, m in (('real map', lo, mask), ('all absent (pad row only)', lo, zero_mask), ('centre only', lo_self, centre),
                   ('27 x rows r..r+2 (perfectly coalesced)', lo_near, full)):
    t8 = timeit(lambda: ops.spconv_cmap(x[1:], l, m, R, w, b, out=out))
    t4 = timeit(lambda: ops.spconv_cmap(x[1:], l, m, R, w4, b4, out=out4))
    print('%-42s  8->8 %.1f us   4->4 (ld 8) %.1f us' % (name, t8, t4))
