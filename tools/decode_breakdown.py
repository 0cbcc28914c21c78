"""Where one decoded loot10 frame spends its time, scale by scale (one linr_decode_scale call each: kernel map, 8 decode stages with
their range decoding on the host, upper_layer)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import codec, overfit, synthetic
from linr_pcgc_amd.module_utils import unique_sorted
from linr_pcgc_amd.model_codec import Model_Estimate
torch.set_num_threads(4)
clouds = [synthetic.sequence_frame_device('loot10', 0, 'cuda')]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
side = dict(enc['side_info']); side.pop('arith_version', None); side['final_bytes'] = enc['model_bin']
m, _ = Model_Estimate().decompress_model(overfit.gen_model(gop.scale_num, 'cuda'), side)
lows, mins = codec.dec_all_frame_low_xyz(enc['low_enc_bytes'])
for rep in range(3):
    xyz_low = torch.tensor(lows[0].astype(np.int32), device='cuda')
    lowx = unique_sorted(xyz_low)
    bits = max(1, int(xyz_low.max()).bit_length())
    fb = list(enc['frames'][0])
    per = []
    torch.cuda.synchronize(); t_all = time.time()
    for s_idx in range(len(fb) - 1, -1, -1):
        bits += 1
        t0 = time.time()
        n = lowx.shape[0]
        lowx = m.decode_scale(lowx, s_idx, fb[s_idx], bits)
        per.append((s_idx, n, round((time.time() - t0) * 1e3, 2)))
    torch.cuda.synchronize()
    print('total %.1f ms; (scale, rows, ms): %s' % ((time.time() - t_all) * 1e3, per))
