"""Wall-clock of the per-frame data preparation (octree levels, occupancy, kernel map) on loot10 frames."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import engine, synthetic                     # noqa: E402
from linr_pcgc_amd.module_utils import prepare_frame           # noqa: E402

for t in range(4):
    pts = synthetic.sequence_frame('loot10', t)
    torch.cuda.synchronize()
    t0 = time.time()
    fr = prepare_frame(pts, None, 64, device='cuda')
    torch.cuda.synchronize()
    t1 = time.time()
    f = engine.Frame(fr['all_input_info'], fr['scale_num'], 'cuda', validate=True, with_arena=False)
    torch.cuda.synchronize()
    t2 = time.time()
    print('frame %d: prepare_frame %.1f ms, Frame (kernel maps) %.1f ms, rows %d' % (t, (t1 - t0) * 1e3, (t2 - t1) * 1e3, f.rows))
for t in range(4, 7):
    pts = synthetic.sequence_frame('loot10', t)
    torch.cuda.synchronize()
    t0 = time.time()
    fr = prepare_frame(pts, None, 64, device='cuda', with_offsets=False)
    f = engine.Frame(fr['all_input_info'], fr['scale_num'], 'cuda', validate=True, with_arena=False)
    torch.cuda.synchronize()
    print('frame %d: prepare_frame(with_offsets=False) + Frame %.1f ms' % (t, (time.time() - t0) * 1e3))
