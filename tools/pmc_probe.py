"""Launches the dominant kernels a few times on frame 0 of loot10 so rocprofv3 --pmc can attribute counters."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import ops, synthetic, engine
from linr_pcgc_amd.module_utils import prepare_frame
dev = 'cuda'
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev, with_arena=False)
R = f.rows
x = torch.zeros((R + 1, 8), device=dev); x[1:].normal_()
go = torch.zeros((R + 1, 8), device=dev); go[1:].normal_()
out = torch.empty((R, 8), device=dev)
w = torch.randn(27, 8, 8, device=dev) * 0.1
b = torch.zeros(1, 8, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ops.spconv_fwd(x[1:], f.nbr, w, b, out=out, pad_row=True)
    ops.spconv_bwd_data(go[1:], f.nbr, w, out=out, pad_row=True)
    ops.spconv_bwd_weight(x[1:], go[1:], f.nbr[:, :R], 8, 8)
torch.cuda.synchronize()
