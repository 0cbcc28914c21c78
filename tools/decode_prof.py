"""Wall-clock breakdown of decoding one loot10 frame (staged decoder)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import codec, engine, model_core, overfit, synthetic                     # noqa: E402
from linr_pcgc_amd.module_utils import BinaryArithmeticCoding, octree_level_obj, qscTensor   # noqa: E402

pts = synthetic.sequence_frame('loot10', 0)
gop = overfit.Gop(None, [pts], None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
T = {}


def tick(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.time() - t0


for rep in range(3):
    T.clear()
    t_all = time.time()
    dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda')
    torch.cuda.synchronize()
    print('decode_gop (1 frame, incl. model decompression): %.3f s' % (time.time() - t_all))

# instrumented copy of the per-frame loop (codec.decode_one_frame + LINR_PCGC_Model.decode_frame)
import numpy as np                                         # noqa: E402
from linr_pcgc_amd import _lib                             # noqa: E402
from linr_pcgc_amd.model_codec import Model_Estimate       # noqa: E402
from linr_pcgc_amd.function_utils import unpack_bitstream  # noqa: E402
from linr_pcgc_amd.module_utils import unique_sorted       # noqa: E402
side = dict(enc['side_info']); side['final_bytes'] = enc['model_bin']
t0 = time.time(); m, _ = Model_Estimate().decompress_model(overfit.gen_model(gop.scale_num, 'cuda'), side); tick('model decompress', t0)
lows, mins = codec.dec_all_frame_low_xyz(enc['low_enc_bytes'])
lowx = unique_sorted(torch.tensor(lows[0].astype('int32'), device='cuda'))
fb = list(enc['frames'][0])
L = _lib.lib()
for s_idx in range(len(fb) - 1, -1, -1):
    t0 = time.time(); streams = unpack_bitstream(fb[s_idx])
    frame = m._scale_frame({'coord': lowx, 'offset_tensor': None, 'scale_idx': s_idx}, need_occ=False)
    tick('frame build (kmap, offsets, arena)', t0)
    frame.occ.zero_()
    rows = frame.rows
    probs = torch.empty((8, rows), dtype=torch.float32, device='cuda')
    p_host, s_host = m._host_buffers(rows)
    for k in range(8):
        t0 = time.time(); engine.net_forward(frame, m._flat, k, k + 1, probs, None); tick('gpu stage forward', t0)
        t0 = time.time(); p_host[:rows].copy_(probs[k]); tick('d2h', t0)
        t0 = time.time(); buf = np.frombuffer(streams[k], dtype=np.uint8)
        L.linr_ac_decode_binary(p_host.numpy().ctypes.data, rows, buf.ctypes.data, buf.size, s_host.numpy().ctypes.data); tick('ac decode', t0)
        t0 = time.time(); frame.occ[:, k] = s_host[:rows].to('cuda').to(torch.float32); tick('h2d', t0)
    t0 = time.time(); lowx = octree_level_obj.upper_layer(lowx, frame.occ.clone()); tick('upper_layer', t0)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print('%-36s %.1f ms' % (k, v * 1e3))
print('sum %.1f ms' % (sum(T.values()) * 1e3))

# GOP-level decode throughput: 8 frames, serial vs threaded
clouds = [synthetic.sequence_frame('loot10', t) for t in range(8)]
gop8 = overfit.Gop(None, clouds, None, 64, 'cuda')
enc8 = codec.encode_gop(model, overfit.gen_model(gop8.scale_num, 'cuda'), gop8, 8)
for w in (1, 2, 4, 8):
    torch.cuda.synchronize(); t0 = time.time()
    out = codec.decode_gop(overfit.gen_model(gop8.scale_num, 'cuda'), enc8, 'cuda', workers=w)
    torch.cuda.synchronize()
    print('decode_gop 8 frames, workers=%d: %.1f ms/frame' % (w, (time.time() - t0) * 1e3 / 8))
