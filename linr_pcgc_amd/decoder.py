"""decoder.py of the reference (decode / decode_one_gop / decode_one_frame, decoder.py:16-176) with its argument dicts: the
files of ``encoder.encode`` (or ``codec.write_gop``) in, every frame rebuilt from them alone, compared with the test data set
and - with ``write_flag`` - written as ``<result_dec_dir>/frameXXXX.ply``.  The frame window of a GOP is inclusive (see
encoder.py here for the reference's off-by-one)."""
import os

import torch

from . import codec
from .codec import decode_one_frame          # noqa: F401  (decoder.py:153-176: the drivers import it from here)
from .custom_dataset import Read_Data_with_cache, write_ply_ascii
from .encoder import gop_bounds


def decode(inargs):
    """decoder.py:16-48.  inargs: 'gop_names', 'result_enc_dir', 'result_dec_dir', 'Gen_Model', 'dataset' (a MytestDataset: the
    sorted voxel list of every frame), 'write_flag'."""
    os.makedirs(inargs['result_dec_dir'], exist_ok=True)
    for gop_name in inargs['gop_names']:
        first, last = gop_bounds(gop_name)
        decode_one_gop({'gop_bound': [first, last], 'frame_num': last - first + 1, 'result_enc_dir': inargs['result_enc_dir'],
                        'result_dec_dir': inargs['result_dec_dir'], 'Gen_Model': inargs['Gen_Model'], 'gop_name': gop_name,
                        'reading_data': Read_Data_with_cache(inargs['dataset'], list(range(first, last + 1))),
                        'write_flag': inargs['write_flag']})


def decode_one_gop(inargs):
    """decoder.py:51-146: model.bin -> parameters, then every frame from its streams and the coarsest coordinates; raises
    AssertionError on the first frame that differs from the data set's."""
    enc = codec.read_gop(os.path.join(inargs['result_enc_dir'], inargs['gop_name']))
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    decoded = codec.decode_gop(inargs['Gen_Model'](), enc, dev, frames=list(range(inargs['frame_num'])), workers=1)
    for frame_idx, dec in enumerate(decoded):
        truth = inargs['reading_data'][frame_idx]
        if dec.shape != truth.shape or bool((dec.to(truth.dtype) != truth).any()):
            raise AssertionError('frame %d of %s does not decode to the input' % (frame_idx, inargs['gop_name']))
        if inargs['write_flag']:
            name = 'frame%s.ply' % str(inargs['gop_bound'][0] + frame_idx).zfill(4)
            write_ply_ascii(os.path.join(inargs['result_dec_dir'], name), dec.cpu().numpy())
    return decoded


def main(argv=None):
    """python -m linr_pcgc_amd.decoder --enc-dir OUT/result_enc --dec-dir OUT/dec [--ori-dir frames --ori-type ply]
    The decoder as its own program (decoder.py:179-200): every GOP under --enc-dir from its files alone, frames written as
    frameXXXX.ply; with --ori-dir each one is also compared with the input.  The model's shape travels in side_info.json (the
    reference hard-codes it, decoder.py:189); for streams without it the scale count is read off the stream files and width /
    block_layers are flags."""
    import argparse
    from .model_core import LINR_PCGC_Model
    ap = argparse.ArgumentParser('linr_pcgc_amd.decoder')
    ap.add_argument('--enc-dir', required=True)
    ap.add_argument('--dec-dir', required=True)
    ap.add_argument('--ori-dir', default=None)
    ap.add_argument('--ori-type', default='ply', choices=['ply', 'npy'])
    ap.add_argument('--hidden-channel-conv', type=int, default=8)
    ap.add_argument('--block-layers', type=int, default=1)
    args = ap.parse_args(argv)
    names = sorted((n for n in os.listdir(args.enc_dir) if n.startswith('gop_')), key=lambda n: gop_bounds(n)[0])
    if not names:
        raise ValueError('no gop_* directory under %s' % args.enc_dir)
    os.makedirs(args.dec_dir, exist_ok=True)
    truth = None
    if args.ori_dir is not None:
        from .custom_dataset import MytestDataset
        truth = MytestDataset(args.ori_dir, ori_type=args.ori_type)
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    frames = 0
    for name in names:
        first, last = gop_bounds(name)
        enc = codec.read_gop(os.path.join(args.enc_dir, name))
        side = enc['side_info']          # streams of this package carry the model's shape; others: the flags, and the most scales a frame has
        shape = {'scale_num': int(side.get('scale_num', max(len(f) for f in enc['frames']))), 'in_channel': 7,
                 'hidden_channel_conv': int(side.get('hidden_channel_conv', args.hidden_channel_conv)),
                 'block_layers': int(side.get('block_layers', args.block_layers)), 'outstage': 8, 'instage': 1}
        gen = lambda: LINR_PCGC_Model(shape).to(dev)
        decoded = codec.decode_gop(gen(), enc, dev, workers=1 if shape['hidden_channel_conv'] != 8 else 4)
        if len(decoded) != last - first + 1:
            raise ValueError('%s holds %d frames, its name says %d' % (name, len(decoded), last - first + 1))
        for i, dec in enumerate(decoded):
            if truth is not None:
                want = torch.unique(truth[first + i], dim=0)
                if dec.shape != want.shape or bool((dec.to(want.dtype) != want).any()):
                    raise AssertionError('frame %d does not decode to the input' % (first + i))
            write_ply_ascii(os.path.join(args.dec_dir, 'frame%s.ply' % str(first + i).zfill(4)), dec.cpu().numpy())
            frames += 1
    print('decoded %d frames of %d GOPs into %s%s' % (frames, len(names), args.dec_dir, '' if truth is None else ' (all equal to the input)'))


if __name__ == '__main__':
    main()
