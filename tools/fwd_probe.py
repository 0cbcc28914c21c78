"""A few executor forward/train steps on frame 0 of loot10 for rocprofv3 --pmc attribution."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import engine, synthetic, overfit
from linr_pcgc_amd.module_utils import prepare_frame
dev = 'cuda'
fr = prepare_frame(synthetic.sequence_frame('loot10', 0), None, 64, device=dev)
model = overfit.gen_model(fr['scale_num'], dev, seed=1)
f = engine.Frame(fr['all_input_info'], fr['scale_num'], dev)
bits = torch.zeros(1, dtype=torch.float64, device=dev)
flat = model.flat_parameters()
m = torch.zeros_like(flat); v = torch.zeros_like(flat)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    engine.net_train_step(f, flat, m, v, 1e-6, 1, 0.0, 0.9, 0.999, 1e-8, 0.0, bits)
torch.cuda.synchronize()
