"""Writes tests/golden/stream_v<ARITH_VERSION>.npz: the coded streams of a tiny fixed cloud under a fixed-seed (untrained) model,
fp32 and bf16, with the coordinates they must decode to.  Run on a GPU box whenever codec.ARITH_VERSION is bumped on purpose:
    gpurun -- 'python tests/golden/make_stream_golden.py gpurun_out/stream_golden.npz'   then copy it to tests/golden/.
tests/test_gpu_drivers.py::test_committed_stream_still_decodes decodes the committed file: a change of the forward's fp32
evaluation order WITHOUT a version bump turns the decoded geometry into garbage there."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from linr_pcgc_amd import codec, overfit, synthetic           # noqa: E402

out = sys.argv[1]
clouds = [synthetic.sphere_shell(6, 20), synthetic.sphere_shell(6, 23)]
gop = overfit.Gop(None, clouds, None, 64, 'cuda')
model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
data = {'arith_version': codec.ARITH_VERSION, 'scale_num': gop.scale_num}
for prec in ('f32', 'bf16'):
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8, precision=prec)
    data[prec + '_model_bin'] = np.frombuffer(enc['model_bin'], dtype=np.uint8)
    data[prec + '_low'] = np.frombuffer(enc['low_enc_bytes'], dtype=np.uint8)
    data[prec + '_side'] = np.array(repr(enc['side_info']))
    for fi, scales in enumerate(enc['frames']):
        for si, b in enumerate(scales):
            data['%s_f%d_s%d' % (prec, fi, si)] = np.frombuffer(b, dtype=np.uint8)
for i in range(2):
    data['ref%d' % i] = (torch.as_tensor(gop.infos[i]['ori']).cpu() + torch.as_tensor(gop.coord_mins[i]).cpu().to(torch.int32)).numpy()
np.savez_compressed(out, **data)
print('wrote', out, os.path.getsize(out), 'bytes')
