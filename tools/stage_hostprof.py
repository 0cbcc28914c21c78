"""Host-side profile (cProfile, cumulative) of the per-frame staging calls on loot10 frames - where the 1 ms per frame goes on the CPU side.
   python tools/stage_hostprof.py [frames]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import engine, synthetic                   # noqa: E402
from linr_pcgc_amd.module_utils import prepare_frame          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pts = [synthetic.sequence_frame_device('loot10', i, 'cuda') for i in range(n)]
for p in pts[:4]:                                              # lazy loads, allocator warm-up
    fr = prepare_frame(p, None, 64, device='cuda', with_offsets=False)
    engine.Frame(fr['all_input_info'], fr['scale_num'], 'cuda', validate=True, with_arena=False)
torch.cuda.synchronize()


def run():
    for p in pts:
        fr = prepare_frame(p, None, 64, device='cuda', with_offsets=False)
        torch.cuda.synchronize()
        engine.Frame(fr['all_input_info'], fr['scale_num'], 'cuda', validate=True, with_arena=False)
        torch.cuda.synchronize()


pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
