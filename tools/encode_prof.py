"""Wall-clock breakdown of the codec leg of bench.py (encode_gop + write_gop) on the default workload:
gpurun -- 'python tools/encode_prof.py'"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import codec, overfit, synthetic                        # noqa: E402
from linr_pcgc_amd.model_codec import Model_Estimate                      # noqa: E402
from linr_pcgc_amd.model_core import FlatAdam, encode_streams             # noqa: E402
from linr_pcgc_amd.function_utils import pack_bitstream                   # noqa: E402

torch.set_num_threads(16)
gop = overfit.Gop(None, [synthetic.sequence_frame_device('loot10', t, 'cuda') for t in range(8)], None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
overfit.overfit_gop(model, FlatAdam(model), gop, 3)
torch.cuda.synchronize()


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n, out


dt, comp = t(lambda: Model_Estimate().compress_model(model, 8, True, overfit.gen_model(gop.scale_num, 'cuda')))
print('compress_model (once per GOP)      %.2f ms' % (dt * 1e3))
coded = comp['new_model']
f = gop.frames[0]
dt, (probs, bits) = t(lambda: coded.frame_probs(f))
print('forward                            %.2f ms' % (dt * 1e3))
dt, p_host = t(lambda: probs.cpu().numpy())
print('D2H probs fp32 (pageable .cpu())   %.2f ms  (%.1f MB)' % (dt * 1e3, probs.numel() * 4 / 1e6))
pin = torch.empty(probs.shape, dtype=torch.float32, pin_memory=True)
dt, _ = t(lambda: pin.copy_(probs))
print('D2H probs fp32 into pinned         %.2f ms' % (dt * 1e3))
dt, occ_host = t(lambda: f.occ.t().to(torch.uint8).contiguous().cpu().numpy())
print('occ transpose + cast + D2H         %.2f ms' % (dt * 1e3))
ps, ss = [], []
for i in range(f.n_scales):
    a, b = int(f.row_off[i]), int(f.row_off[i + 1])
    for k in range(8):
        ps.append(p_host[k, a:b]); ss.append(occ_host[k, a:b])
for nt in (8, 16):
    dt, streams = t(lambda: encode_streams(ps, ss, nt))
    print('range coding, %2d threads           %.2f ms  (%d symbols, %d bytes)' % (nt, dt * 1e3, sum(p.size for p in ps), sum(len(s) for s in streams)))
dt, _ = t(lambda: [pack_bitstream(streams[8 * i:8 * i + 8]) for i in range(f.n_scales)])
print('pack_bitstream                     %.2f ms' % (dt * 1e3))
dt, enc = t(lambda: codec.encode_gop(model, overfit.gen_model(gop.scale_num, "cuda"), gop, 8), 2)
print('encode_gop, 8 frames               %.2f ms/frame' % (dt * 1e3 / 8))
import tempfile
d = tempfile.mkdtemp()
dt, _ = t(lambda: codec.write_gop(enc, d), 2)
print('write_gop, 8 frames                %.2f ms/frame' % (dt * 1e3 / 8))
