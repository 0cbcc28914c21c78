"""Generates the committed golden fixtures from the reference itself.  Runs ONLY in the build container
(/root/reference is read-only there and absent on the GPU box); the outputs in this directory are data.

What is executed is the reference's own pure-torch code: models/module_utils.py (qscTensor, QuickSearchCoord,
octree_level.forward/upper_layer), models/sort_functions.py, models/quantize_functions.py,
models/function_utils.py (pack_bitstream / unpack_bitstream).  MinkowskiEngine / torchac / open3d are absent
from this image, so empty placeholder modules satisfy the import statements; no placeholder code runs in any
function called below.  `Tensor.cuda` is made an identity because the reference hard-codes `.cuda()`
(module_utils.py:93) and this container has no GPU.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    me = types.ModuleType('MinkowskiEngine')

    class _Absent:
        def __init__(self, *a, **k):
            raise RuntimeError('MinkowskiEngine is not available')
    me.SparseTensor = _Absent
    me.MinkowskiPruning = lambda *a, **k: None
    sys.modules['MinkowskiEngine'] = me
    sys.modules['torchac'] = types.ModuleType('torchac')
    sys.modules['open3d'] = types.ModuleType('open3d')
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'models'))
    import models.module_utils as mu
    import models.function_utils as fu
    from glob_params import offsets_ini
    return mu, fu, offsets_ini


def octree_fixture(mu, offsets_ini, points, min_point_num, name):
    """Same loop as MyDataset.handle_data (custom_dataset.py:259-355), calling the reference's helpers."""
    pts = points[:, :3]
    cmin = pts.min(axis=0)
    xyz = torch.unique(torch.tensor(pts - cmin, dtype=torch.int32), dim=0)
    cur = mu.qscTensor(xyz, torch.ones((xyz.shape[0], 1)))
    out = {'points': points.astype(np.int32), 'coord_data_min': cmin.astype(np.int32),
           'ori': cur.get_coord().numpy(), 'min_point_num': np.int32(min_point_num)}
    s = 0
    while True:
        cur.set_oct_level()
        parent, occ = cur.get_oct_level()
        rec = mu.octree_level_obj.upper_layer(parent, occ)
        assert (rec != cur.coord).sum() == 0
        low = mu.qscTensor(parent, torch.ones((parent.shape[0], 1)))
        low.set_offset_tensor(offsets_ini)
        out['s%d_coord' % s] = low.get_coord().numpy().astype(np.int32)
        out['s%d_occ' % s] = occ.numpy().astype(np.float32)
        out['s%d_offset' % s] = low.get_offset_tensor().numpy().astype(np.float32)
        out['s%d_upper' % s] = rec.numpy().astype(np.int32)
        if parent.shape[0] < min_point_num:
            break
        cur = low
        s += 1
    out['scale_num'] = np.int32(s + 1)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'scales', s + 1, 'points', xyz.shape[0])


def main():
    mu, fu, offsets_ini = import_reference()
    rng = np.random.default_rng(20240607)

    # (1) ragged random cloud with duplicates and a non-zero minimum
    pts = rng.integers(0, 64, size=(6000, 3)) + np.array([5, 17, 3])
    octree_fixture(mu, offsets_ini, pts, 64, 'octree_random64.npz')
    # (2) thin surface: quarter sphere shell, 7 bit
    g = np.stack(np.meshgrid(*[np.arange(128)] * 3, indexing='ij'), -1).reshape(-1, 3)
    d = np.sqrt(((g - 64) ** 2).sum(1))
    shell = g[(np.abs(d - 50) < 0.5) & (g[:, 0] >= 64)]
    octree_fixture(mu, offsets_ini, shell, 64, 'octree_shell128.npz')

    # (3) bitstream packing (function_utils.py:109-132)
    streams = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in (0, 1, 7, 300, 4096)]
    packed = fu.pack_bitstream(streams)
    assert [bytes(b) for b in fu.unpack_bitstream(packed)] == streams
    np.savez_compressed(os.path.join(HERE, 'pack_bitstream.npz'),
                        packed=np.frombuffer(packed, dtype=np.uint8),
                        lens=np.array([len(s) for s in streams], dtype=np.int64),
                        payload=np.frombuffer(b''.join(streams), dtype=np.uint8))

    # (4) weight-quantiser / model-stream known answers from the shipped run artefacts (data files)
    ck = torch.load(os.path.join(REF, 'loot/gop_32_62/model.pth'), map_location='cpu', weights_only=True)
    names = list(ck['model'].keys())
    flat = torch.cat([v.reshape(-1) for v in ck['model'].values()]).numpy()
    side = json.load(open(os.path.join(REF, 'loot/gop_32_62/70/side_info.json')))
    res = json.load(open(os.path.join(REF, 'loot/gop_32_62/70/result.json')))
    np.savez_compressed(os.path.join(HERE, 'loot_model_kat.npz'), flat=flat.astype(np.float32),
                        names=np.array(names), shapes=np.array([str(tuple(v.shape)) for v in ck['model'].values()]),
                        mu=side['mu'], b=side['b'], min_param=side['min_param'], max_param=side['max_param'],
                        model_bpp=res['model_bpp'], bpp_t=res['bpp_t'], xyzlow_bpp=res['xyzlow_bpp'],
                        epoch=ck['epoch'], bitdepth=ck['bitdepth'])
    print('loot_model_kat.npz', flat.shape, side)


if __name__ == '__main__':
    main()
