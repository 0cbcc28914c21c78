"""ms per training step of the channel-blocked executor (hidden_channel_conv 16 / 32) on frame 0 of a config, beside width 8."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit, synthetic
from linr_pcgc_amd.model_core import FlatAdam, train_step
cfg = sys.argv[1] if len(sys.argv) > 1 else 'loot10'
gop = overfit.Gop(None, [synthetic.sequence_frame_device(cfg, 0, 'cuda')], None, 64, 'cuda')
for hidden in (8, 16, 32):
    m = overfit.gen_model(gop.scale_num, 'cuda', seed=8807, hidden=hidden)
    o = FlatAdam(m)
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    for _ in range(3):
        train_step(m, o, gop.frames[0], gop.point_nums[0], out=bits)
    torch.cuda.synchronize(); t0 = time.time(); n = 10
    first = None
    for i in range(n):
        bits.zero_(); train_step(m, o, gop.frames[0], gop.point_nums[0], out=bits)
        if first is None: first = float(bits)
    torch.cuda.synchronize()
    print('%s hidden %2d: %.2f ms/step, %d parameters, bits %.0f -> %.0f' % (cfg, hidden, (time.time() - t0) * 1e3 / n, m.flat_parameters().numel(), first, float(bits)))
