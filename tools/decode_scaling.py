"""Decode throughput of one loot10 GOP against the number of frames in flight (codec.decode_gop workers); the model is trained
first (argv[2] epochs, default 20) so that the streams have the entropy of real ones - an untrained model's ~1 bit/symbol
streams cost the range decoder three times as long."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import codec, overfit, synthetic                     # noqa: E402
from linr_pcgc_amd.model_core import FlatAdam                           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.set_num_threads(4)
clouds = [synthetic.sequence_frame_device('loot10', t, 'cuda') for t in range(n)]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if epochs:
    print('trained %d epochs: %.3f bpp' % (epochs, min(overfit.overfit_gop(model, FlatAdam(model), gop, epochs))))
enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
for workers in (1, 2, 4, 8, 12, 16):
    if workers > n:
        break
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', workers=workers)
        torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    print('workers %2d: %.1f ms per frame (%d frames, %d host threads available)' % (workers, best * 1e3 / n, n, len(os.sched_getaffinity(0))))
