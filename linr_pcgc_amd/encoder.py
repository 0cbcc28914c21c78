"""encoder.py of the reference (encode / encode_one_gop / encode_one_frame, encoder.py:13-176) with its argument dicts, on the
HIP-backed model: a checkpoint per GOP in, ``<encode_dir>/<gop>/bins/{frameXXXX_scaleY.bin, model.bin, low_enc_bytes.bin}`` +
``side_info.json`` out.  This is the reference-shaped road (one model.encode call per scale and frame); ``codec.encode_gop`` is
the batched, pipelined one the sequence driver uses - both write the same files.

One deliberate difference: the reference builds a GOP's frame window as ``range(first, last)`` (encoder.py:47), which drops the
last frame and fails on a one-frame GOP; the window here is inclusive, like the GOP names.
"""
import json
import os

import torch

from .custom_dataset import Read_Data_with_cache
from .model_codec import Model_Estimate
from .test_utils import enc_all_frame_low_xyz, write_bin_file
from . import codec


def gop_bounds(gop_name):
    first, last = (int(d) for d in gop_name.replace('gop_', '').split('_'))
    return first, last


def encode(enc_args):
    """encoder.py:20-54.  enc_args: 'outputdir' (holds <gop>/model.pth), 'gop_names', 'Gen_Model', 'dataset', 'encode_dir'."""
    os.makedirs(enc_args['encode_dir'], exist_ok=True)
    for gop_name in enc_args['gop_names']:
        first, last = gop_bounds(gop_name)
        reading = Read_Data_with_cache(enc_args['dataset'], list(range(first, last + 1)))
        encode_one_gop({'Gen_Model': enc_args['Gen_Model'], 'frame_num': last - first + 1,
                        'model_path': os.path.join(enc_args['outputdir'], gop_name, 'model.pth'),
                        'result_dir': os.path.join(enc_args['encode_dir'], gop_name), 'reading_data': reading,
                        'low_enc_bytes': enc_all_frame_low_xyz(reading, last - first + 1)})


def encode_one_frame(model, all_inargs, ori=None):
    """encoder.py:158-176: model.encode on every scale of a frame."""
    all_bit, all_bytes = 0, []
    for inargs in all_inargs:
        putin = dict(inargs)
        qsc = inargs['xyzqsc_t']
        putin['coord'], putin['offset_tensor'] = qsc.get_coord(), qsc.get_offset_tensor()
        ret = model.encode(putin)
        all_bit += ret['bits']
        all_bytes.append(ret['enc_bytes'])
    return {'all_bit': all_bit, 'all_bytes': all_bytes}


def encode_one_gop(inargs):
    """encoder.py:57-156.  Returns {'bpp_all', 'point_bpp', 'model_bpp', 'xyzlow_bpp'} (test_utils.py:146-157's accounting; the
    two extra side-info fields of this codec counted with the model)."""
    low = inargs.get('low_enc_bytes')
    if low is None:
        raise ValueError('low_enc_bytes is None')
    gen = inargs['Gen_Model']
    ckpt = torch.load(inargs['model_path'], map_location='cpu', weights_only=False)
    trained = gen()
    trained.load_state_dict(ckpt['model'])
    bitdepth = ckpt.get('bitdepth') or 8
    result_dir = inargs['result_dir']
    bins_dir = os.path.join(result_dir, 'bins')
    os.makedirs(bins_dir, exist_ok=True)
    with open(os.path.join(bins_dir, 'low_enc_bytes.bin'), 'wb') as f:
        f.write(low)
    packed = Model_Estimate().compress_model(trained, bitdepth, True, gen())
    with open(os.path.join(bins_dir, 'model.bin'), 'wb') as f:
        f.write(packed['final_bytes'])
    side_info = {'mu': packed['mu'], 'b': packed['b'], 'min_param': packed['min_param'], 'max_param': packed['max_param'],
                 'enc_mode': packed['enc_mode'], 'bitdepth': bitdepth, 'arith_version': codec.ARITH_VERSION}
    side_info.update(codec.model_shape(trained))
    with open(os.path.join(result_dir, 'side_info.json'), 'w') as f:
        json.dump(side_info, f, indent=4)
    model = packed['new_model']
    bits, points = 0, 0
    for frame_idx in range(inargs['frame_num']):
        frame = inargs['reading_data'][frame_idx]
        out = encode_one_frame(model, frame['all_input_info'], frame.get('ori'))
        write_bin_file(frame_idx, out['all_bytes'], bins_dir)
        bits += out['all_bit']
        points += frame['point_num']
    model_bits = packed['bit_real'] + codec.EXTRA_SIDE_BITS
    return {'point_bpp': bits / points, 'model_bpp': model_bits / points, 'xyzlow_bpp': len(low) * 8 / points,
            'bpp_all': (bits + model_bits + len(low) * 8) / points}
