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
