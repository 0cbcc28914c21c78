"""Where one decoded loot10 frame spends its time: stream unpacking, frame construction, the C stage loop, upper_layer."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import codec, overfit, synthetic, model_core, engine
from linr_pcgc_amd.function_utils import unpack_bitstream
from linr_pcgc_amd.module_utils import octree_level_obj, unique_sorted
from linr_pcgc_amd.model_codec import Model_Estimate
import numpy as np
torch.set_num_threads(4)
clouds = [synthetic.sequence_frame_device('loot10', 0, 'cuda')]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
side = dict(enc['side_info']); side.pop('arith_version', None); side['final_bytes'] = enc['model_bin']
m, _ = Model_Estimate().decompress_model(overfit.gen_model(gop.scale_num, 'cuda'), side)
lows, mins = codec.dec_all_frame_low_xyz(enc['low_enc_bytes'])
for rep in range(3):
    T = {}
    def tick(n, t0):
        torch.cuda.synchronize(); T[n] = T.get(n, 0) + time.time() - t0
    lowx = unique_sorted(torch.tensor(lows[0].astype('int32'), device='cuda'))
    fb = list(enc['frames'][0])
    t_all = time.time()
    for s_idx in range(len(fb) - 1, -1, -1):
        t0 = time.time(); streams = unpack_bitstream(fb[s_idx]); tick('unpack', t0)
        t0 = time.time(); frame = m.make_frame([{'coord': lowx, 'offset_tensor': None, 'scale_idx': s_idx}], with_arena=True); tick('make_frame', t0)
        t0 = time.time(); occ = m.decode_frame(frame, [streams]); tick('decode_frame (C loop)', t0)
        t0 = time.time(); lowx = octree_level_obj.upper_layer(lowx, torch.cat(occ, dim=-1)); tick('upper_layer', t0)
    torch.cuda.synchronize()
    print('total %.1f ms' % ((time.time() - t_all) * 1e3), {k: round(v * 1e3, 1) for k, v in T.items()})
