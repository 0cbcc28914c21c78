"""N training steps of one executor on loot10 frames, nothing else: the program rocprofv3 profiles (tools/prof_bf16.sh).
  python3 tools/bf16_steps.py [bf16|f32] [steps] [config]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linr_pcgc_amd import overfit, synthetic                    # noqa: E402
from linr_pcgc_amd.model_core import FlatAdam, train_step       # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
config = sys.argv[3] if len(sys.argv) > 3 else 'loot10'
gop = overfit.Gop(None, [synthetic.sequence_frame_device(config, t, 'cuda') for t in range(4)], None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
model.train_precision = prec
opt = FlatAdam(model)
bits = torch.zeros(1, dtype=torch.float64, device='cuda')
for i in range(steps):
    train_step(model, opt, gop.frames[i % 4], gop.point_nums[i % 4], out=bits)
torch.cuda.synchronize()
print('steps', steps, 'rows', [f.rows for f in gop.frames], 'bits', float(bits))
