"""Runs a few training steps of the executor on frame 0 of loot10 (the grouped launches exactly as bench.py times them) so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can attribute HBM traffic to the kernels of a step (tools/traffic_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit, synthetic                        # noqa: E402
from linr_pcgc_amd.model_core import FlatAdam, train_step          # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
precision = sys.argv[2] if len(sys.argv) > 2 else 'f32'          # 'bf16': the bf16 training executor (linr_net_train_step_bf16)
config = sys.argv[3] if len(sys.argv) > 3 else 'loot10'           # 'loot10_rough': the non-spherical stress workload (tools/rough_pmc.sh)
gop = overfit.Gop(None, [synthetic.sequence_frame_device(config, 0, 'cuda')], None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
model.train_precision = precision
opt = FlatAdam(model)
acc = torch.zeros(1, dtype=torch.float64, device='cuda')
for _ in range(steps):
    train_step(model, opt, gop.frames[0], gop.point_nums[0], out=acc)
torch.cuda.synchronize()
print('rows', gop.frames[0].rows, 'steps', steps, 'precision', precision, 'config', config)
