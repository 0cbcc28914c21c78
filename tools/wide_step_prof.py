"""N training steps of the channel-blocked executor on frame 0 of loot10 - the program the profilers run (tools/wide_counters.sh).
   python tools/wide_step_prof.py [hidden = 16] [steps = 13]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit, synthetic
from linr_pcgc_amd.model_core import FlatAdam, train_step
gop = overfit.Gop(None, [synthetic.sequence_frame_device('loot10', 0, 'cuda')], None, 64, 'cuda')
m = overfit.gen_model(gop.scale_num, 'cuda', seed=8807, hidden=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
o = FlatAdam(m)
bits = torch.zeros(1, dtype=torch.float64, device='cuda')
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 13):
    train_step(m, o, gop.frames[0], gop.point_nums[0], out=bits)
torch.cuda.synchronize()
