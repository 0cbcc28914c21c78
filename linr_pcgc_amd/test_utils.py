"""Mid-test driver of the reference (test_utils.py:16-196): the metrics pass main.py runs on a checkpoint every
``--mid_test`` epoch (main.py:365-380) and the source of the numbers it ships (``loot/gop_32_62/<epoch>/result.json``).

It measures through ``model.codec`` - one forward per scale with the 8 stages coded as ONE stream and decoded again on the spot
(models/model_core.py:169-227) - not through the staged decoder, and writes the reference's files: ``result.json``,
``side_info.json`` and, with ``write_flag``, ``bins/frameXXXX_scaleY.bin``, ``bins/model.bin``, ``bins/low_enc_bytes.bin``.
Same argument dict, same result keys; the model underneath is the HIP-backed ``LINR_PCGC_Model``.  (The streams written here
hold one stream per scale, like the reference's mid-test; the per-stage streams a decoder can consume come from
``codec.encode_gop`` / ``model.encode``.)
"""
import json
import os

import torch


def write_bin_file(frame_idx, all_bytes, bins_dir):
    """test_utils.py:299-305: bins/frameXXXX_scaleY.bin, one file per scale."""
    for scale_idx, payload in enumerate(all_bytes):
        with open(os.path.join(bins_dir, 'frame%s_scale%d.bin' % (str(frame_idx).zfill(4), scale_idx)), 'wb') as f:
            f.write(payload)


def test_one_frame(model, all_inargs):
    """test_utils.py:166-196: model.codec on every scale of a frame; real bits, training-loss bits, payloads and times."""
    out = {'all_bit': 0, 'all_bit_t': 0, 'all_bytes': [], 'enc_time': 0.0, 'dec_time': 0.0}
    for inargs in all_inargs:
        putin = dict(inargs)
        qsc = inargs['xyzqsc_t']
        putin['coord'], putin['offset_tensor'] = qsc.get_coord(), qsc.get_offset_tensor()
        ret = model.codec(putin)
        out['all_bit'] += ret['bits']
        out['all_bit_t'] = out['all_bit_t'] + ret['bits_t']
        out['all_bytes'].append(ret['enc_bytes'])
        out['enc_time'] += ret['enc_time']
        out['dec_time'] += ret['dec_time']
    return out


test_one_frame.__test__ = False          # a driver function of the reference's name, not a pytest case


def Test_one_gop(inargs):
    """test_utils.py:16-163.  inargs: 'model_path' (checkpoint with 'model' and optionally 'bitdepth'), 'Gen_Model' (factory),
    'frame_num', 'compress_model_test' (Model_Estimate().compress_test), 'reading_data' (indexable: frame dicts with
    'all_input_info' and 'point_num'), 'result_dir', 'write_flag', 'low_enc_ret' (bytes of enc_all_frame_low_xyz; required with
    write_flag).  Returns and writes {'bpp_all', 'point_bpp', 'point_bpp_val', 'model_bpp', 'xyzlow_bpp', 'enc_mode', 'enc_time',
    'dec_time'} - the times per frame, the model's compression time included, as in the reference."""
    write_flag, gen_model, frame_num = inargs['write_flag'], inargs['Gen_Model'], inargs['frame_num']
    low_enc_ret = inargs.get('low_enc_ret')
    if low_enc_ret is None and write_flag:
        raise ValueError('low_enc_ret is None while write_flag is True')
    ckpt = torch.load(inargs['model_path'], map_location='cpu', weights_only=False)
    trained = gen_model()
    trained.load_state_dict(ckpt['model'])
    bitdepth = ckpt.get('bitdepth') or 8
    result_dir = inargs['result_dir']
    bins_dir = os.path.join(result_dir, 'bins')
    os.makedirs(bins_dir, exist_ok=True)
    if write_flag:
        with open(os.path.join(bins_dir, 'low_enc_bytes.bin'), 'wb') as f:
            f.write(low_enc_ret)
    # the model goes through its codec first: every frame is coded with the de-quantised parameters the decoder will have
    packed = inargs['compress_model_test'](trained, gen_model(), bitdepth)
    model = packed['new_model']
    enc_time, dec_time = packed['enc_time'], packed['dec_time']
    if write_flag:
        with open(os.path.join(bins_dir, 'model.bin'), 'wb') as f:
            f.write(packed['final_bytes'])
    scalar = lambda v: v.item() if hasattr(v, 'item') else v
    side_info = {'mu': scalar(packed['mu']), 'b': scalar(packed['b']), 'min_param': scalar(packed['min_param']),
                 'max_param': scalar(packed['max_param']), 'enc_mode': packed['enc_mode'], 'xlow_enc_flags': '', 'xlow_enc_modes': ''}
    with open(os.path.join(result_dir, 'side_info.json'), 'w') as f:
        json.dump(side_info, f, indent=4)
    bits_real, bits_loss, points = 0, 0.0, 0
    reading_data = inargs['reading_data']
    for frame_idx in range(frame_num):
        frame = reading_data[frame_idx]
        got = test_one_frame(model, frame['all_input_info'])
        bits_real += got['all_bit']
        bits_loss = bits_loss + got['all_bit_t']
        points += frame['point_num']
        enc_time += got['enc_time']
        dec_time += got['dec_time']
        if write_flag:
            write_bin_file(frame_idx, got['all_bytes'], bins_dir)
    xyzlow_bits = len(low_enc_ret) * 8 if low_enc_ret is not None else 0
    point_bpp, model_bpp, xyzlow_bpp = bits_real / points, packed['bit_real'] / points, xyzlow_bits / points
    result = {'bpp_all': point_bpp + model_bpp + xyzlow_bpp, 'point_bpp': point_bpp, 'point_bpp_val': float(bits_loss) / points,
              'model_bpp': model_bpp, 'xyzlow_bpp': xyzlow_bpp, 'enc_mode': packed['enc_mode'], 'enc_time': enc_time / frame_num,
              'dec_time': dec_time / frame_num}
    with open(os.path.join(result_dir, 'result.json'), 'w') as f:
        json.dump(result, f, indent=4)
    return result


Test_one_gop.__test__ = False


# ---- the coarsest level and the coordinate minima of a GOP (test_utils.py:199-262,299-312) ------------------------------
def enc_oneframe_lowx(frame_data):
    """test_utils.py:235-255: the coarsest scale's voxel coordinates of one frame as uint8 bytes (they must fit 8 bits)."""
    import numpy as np
    low = frame_data['all_input_info'][-1]['xyzqsc_t'].get_coord()
    if int(np.ceil(np.log2(int(low.max()) + 1))) > 8:
        raise AssertionError('downsampled xyzQ should be less than 8 bit')
    return low.detach().cpu().numpy().astype(np.uint8).tobytes()


def xyzlow_tail_handle(all_coord_data_min, all_xlow_info):
    """test_utils.py:257-262: the per-frame payloads followed by all coordinate minima (int32), packed."""
    import numpy as np
    from .function_utils import pack_bitstream
    mins = np.concatenate([np.asarray(m).reshape(-1) for m in all_coord_data_min], axis=0).astype(np.int32)
    return pack_bitstream(list(all_xlow_info) + [mins.tobytes()])


def enc_all_frame_low_xyz(reading_data, frame_num):
    """test_utils.py:199-232."""
    frames = [reading_data[i] for i in range(frame_num)]
    return xyzlow_tail_handle([f['coord_data_min'] for f in frames], [enc_oneframe_lowx(f) for f in frames])


def dec_all_frame_low_xyz(low_byte):
    """test_utils.py:299-312: {'all_xyz_low': list of uint8 [n, 3] arrays, 'all_coord_data_min': int32 [frames, 3] tensor}."""
    from . import codec
    lows, mins = codec.dec_all_frame_low_xyz(low_byte)
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    return {'all_xyz_low': lows, 'all_coord_data_min': torch.tensor(mins.tolist(), dtype=torch.int32, device=dev).reshape(-1, 3)}
