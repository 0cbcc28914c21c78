"""Where one decoded loot10 frame spends its time, scale by scale (one linr_decode_scale call each: kernel map, 8 decode stages with
their range decoding on the host, upper_layer), with a TRAINED model (argv[1] epochs over a 4-frame GOP, default 40 = ~0.3 bpp)
so that the streams have the entropy of real ones; and the range decoder alone on the same 56 streams and probabilities."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import codec, overfit, synthetic, _lib
from linr_pcgc_amd.model_core import FlatAdam
from linr_pcgc_amd.module_utils import unique_sorted
from linr_pcgc_amd.function_utils import unpack_bitstream
from linr_pcgc_amd.model_codec import Model_Estimate
torch.set_num_threads(4)
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
clouds = [synthetic.sequence_frame_device('loot10', t, 'cuda') for t in range(4)]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
if epochs:
    losses = overfit.overfit_gop(model, FlatAdam(model), gop, epochs)
    print('trained %d epochs: %.3f bpp' % (epochs, min(losses)))
enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
print('bpp', {k: round(float(v), 4) for k, v in enc['bpp'].items()})
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

# the range decoder alone: the encoder's probabilities of frame 0 (bitwise the decoder's), the 56 streams, one after the other
L = _lib.lib()
f0 = gop.frames[0]
probs, _ = m.frame_probs(f0)
p_host = probs.cpu().numpy()
row_off = [int(v) for v in f0.row_off]
out = np.empty(f0.rows, np.uint8)
for rep in range(3):
    t0 = time.perf_counter(); nsym = 0
    for j in range(f0.n_scales):
        s_idx = int(f0.scale_idx[j])
        streams = [np.frombuffer(b, dtype=np.uint8) for b in unpack_bitstream(fb[s_idx])]
        r0, r1 = row_off[j], row_off[j + 1]
        for k in range(8):
            pk = p_host[k, r0:r1]
            rc = L.linr_ac_decode_binary(ctypes.c_void_p(pk.ctypes.data), r1 - r0, ctypes.c_void_p(streams[k].ctypes.data if streams[k].size else None),
                                         int(streams[k].size), ctypes.c_void_p(out.ctypes.data))
            assert rc == 0
            nsym += r1 - r0
    dt = time.perf_counter() - t0
    print('range decoder alone: %d symbols in %.2f ms = %.2f ns/symbol' % (nsym, dt * 1e3, dt / nsym * 1e9))
