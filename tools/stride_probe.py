"""Does the row stride of the gathered matrix matter (16-B rows contiguous vs every other 16 B)?"""
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
w = torch.randn(27, 4, 4, device=dev) * 0.1
b = torch.zeros(1, 4, device=dev)
out = torch.empty((R, 4), device=dev)
x4 = torch.zeros((R + 1, 4), device=dev); x4[1:].normal_()
x8 = torch.zeros((R + 1, 8), device=dev); x8[1:].normal_()
x16 = torch.zeros((R + 1, 16), device=dev); x16[1:].normal_()
for name, x in (('ld=4 (16-B rows contiguous)', x4[1:]), ('ld=8 slice [:,0:4]', x8[1:, 0:4]), ('ld=8 slice [:,4:8]', x8[1:, 4:8]), ('ld=16 slice [:,0:4]', x16[1:, 0:4])):
    print('conv 4->4  %-30s PAD %.1f us   branch %.1f us' % (name, timeit(lambda: ops.spconv_fwd(x, f.nbr, w, b, out=out, pad_row=True)),
                                                               timeit(lambda: ops.spconv_fwd(x, f.nbr, w, b, out=out, pad_row=False))))
