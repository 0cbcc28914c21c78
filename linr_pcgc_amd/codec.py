"""Per-GOP encode / decode: mirror of encoder.py:57-203, decoder.py:51-176 and test_utils.py:199-232,299-312.

Stream layout is the reference's: per frame and scale ``pack_bitstream`` of the 8 per-stage
torchac-format streams (csrc/ac.cpp); ``low_enc_bytes`` = pack of per-frame uint8 coarsest coordinates + int32 minima;
``model.bin`` + side info from model_codec.  An encoded GOP is a dict of byte strings; ``write_gop`` / ``read_gop``
map it to the reference's directory layout (bins/frameFFFF_scaleS.bin, bins/model.bin, bins/low_enc_bytes.bin,
side_info.json).
"""
import glob
import json
import os

import numpy as np
import torch

from .function_utils import pack_bitstream, unpack_bitstream
from .model_codec import Model_Estimate
from .model_core import encode_streams
from .module_utils import octree_level_obj, qscTensor, unique_sorted  # noqa: F401

# The probabilities the range coder sees must be reproduced BIT FOR BIT by the decoder, so the fp32 evaluation order of the
# network is part of the stream format.  ARITH_VERSION names it: 1 = taps in ascending index order (rounds 1 and 2 up to the
# "joined schedule" builds), 2 = slab-major tap order (csrc/common.h: LINR_TAP).  A stream carries the version it was coded
# with in side_info.json; decoding one of another version raises instead of silently producing wrong geometry.
ARITH_VERSION = 2


def enc_all_frame_low_xyz(gop):
    """test_utils.py:199-232: uint8 coarsest-scale coordinates per frame + int32 coordinate minima."""
    chunks = []
    for low in gop.low_xyz:
        mx = int(low.max())
        if int(np.ceil(np.log2(mx + 1))) > 8:
            raise AssertionError('downsampled xyzQ should be less than 8 bit')
        chunks.append(low.cpu().numpy().astype(np.uint8).tobytes())
    chunks.append(np.asarray(gop.coord_mins, dtype=np.int32).reshape(-1).tobytes())
    return pack_bitstream(chunks)


def dec_all_frame_low_xyz(low_byte):
    """test_utils.py:299-312."""
    chunks = unpack_bitstream(low_byte)
    mins = np.frombuffer(chunks.pop(), dtype=np.int32).reshape(-1, 3)
    return [np.frombuffer(c, dtype=np.uint8).reshape(-1, 3) for c in chunks], mins


EXTRA_SIDE_BITS = 40          # arith_version, precision and the model's shape (scale_num, block_layers, hidden_channel_conv: the reference's
                              # decoder hard-codes it, decoder.py:189) in side_info.json, one byte each, beyond the reference's side information


def model_shape(model):
    """What a decoder needs to build the model a stream was coded with (the reference's decoder hard-codes it, decoder.py:189)."""
    return {'scale_num': int(model.scale_num), 'block_layers': int(model.block_layers), 'hidden_channel_conv': int(model.hidden)}


def encode_gop(model, model_ori, gop, bitdepth=8, n_threads=None, precision='f32'):
    """encoder.encode_one_gop: quantise the model, then per frame ONE forward over all scales and 8 x scales
    independent arithmetic-coded streams (thread pool).  precision='bf16': the forward runs on the bf16 / uint8-weight
    executor (BASELINE config[4]); the choice is recorded in side_info so that the decoder reproduces it."""
    if n_threads is None:
        n_threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 8))
    comp = Model_Estimate().compress_model(model, bitdepth, True, model_ori)
    coded_model = comp['new_model']                      # the de-quantised model is what codes the geometry
    coded_model.inference_precision = precision
    side_info = {'mu': comp['mu'], 'b': comp['b'], 'min_param': comp['min_param'], 'max_param': comp['max_param'],
                 'enc_mode': comp['enc_mode'], 'bitdepth': bitdepth}
    if precision != 'f32':
        side_info['precision'] = precision
    side_info['arith_version'] = ARITH_VERSION
    side_info.update(model_shape(model))
    # informational (the decoder does not need it): which training executor produced the coded model
    side_info['train_precision'] = getattr(model, 'train_precision', 'f32')
    # Pipeline: the GPU forward + D2H of frame i+1 runs while host workers range-code earlier frames (the coder's C call
    # releases the GIL).  A frame has 8 x scales independent streams, but the 8 streams of its finest scale carry 73 % of the
    # symbols, so one frame keeps only ~8 threads busy: TWO frames are coded concurrently, each on half of the threads
    # (measured: 2.1 ms per frame with one worker x 16 threads, the forward + copies take 1.3 ms).
    from concurrent.futures import ThreadPoolExecutor
    workers = 2 if n_threads >= 4 else 1
    per_job = max(1, n_threads // workers)

    def code(p_host, occ_host, row_off, n_scales):
        ps, ss = [], []
        for i in range(n_scales):
            a, b = int(row_off[i]), int(row_off[i + 1])
            for k in range(8):
                ps.append(p_host[k, a:b])
                ss.append(occ_host[k, a:b])
        streams = encode_streams(ps, ss, per_job)
        return [pack_bitstream(streams[8 * i:8 * i + 8]) for i in range(n_scales)]

    # Device side: forward of frame i+1 on the caller's stream while a copy stream moves the probabilities (fp32) and the
    # occupancy symbols (cast to 1 byte on the GPU) of frame i into a ring of pinned buffers; the host thread only waits for
    # a copy that is already one frame old before it hands the buffers to a coder worker.
    RING = 4
    max_rows = max(f.rows for f in gop.frames)
    dev = gop.frames[0].device
    p_pin = [torch.empty(8 * max_rows, dtype=torch.float32, pin_memory=True) for _ in range(RING)]
    o_pin = [torch.empty(8 * max_rows, dtype=torch.uint8, pin_memory=True) for _ in range(RING)]
    copy_stream = torch.cuda.Stream(device=dev)
    jobs, bits_dev, pending = [], [], []

    def hand_over(entry, coder):
        slot, rows, f, keep, ev_c = entry
        ev_c.synchronize()
        p_host = p_pin[slot][:8 * rows].view(8, rows).numpy()
        occ_host = o_pin[slot][:8 * rows].view(8, rows).numpy()
        jobs.append(coder.submit(code, p_host, occ_host, f.row_off, f.n_scales))

    with ThreadPoolExecutor(max_workers=workers) as coder:
        for i, f in enumerate(gop.frames):
            slot = i % RING
            if i >= RING:
                jobs[i - RING].result()                     # the worker that read this slot's buffers has finished
            probs, bits = coded_model.frame_probs(f)
            bits_dev.append(bits)
            occ_t = f.occ.t().to(torch.uint8).contiguous()
            ev_f = torch.cuda.Event()
            ev_f.record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev_f)
                p_pin[slot][:8 * f.rows].view(8, f.rows).copy_(probs, non_blocking=True)
                o_pin[slot][:8 * f.rows].view(8, f.rows).copy_(occ_t, non_blocking=True)
                ev_c = torch.cuda.Event()
                ev_c.record(copy_stream)
            pending.append((slot, f.rows, f, (probs, occ_t), ev_c))         # keeps the device tensors alive until copied
            if len(pending) > 1:
                hand_over(pending.pop(0), coder)
        while pending:
            hand_over(pending.pop(0), coder)
        frames_bytes = [j.result() for j in jobs]
    bits_est = float(torch.stack(bits_dev).sum())
    low = enc_all_frame_low_xyz(gop)
    points = sum(gop.point_nums)
    occ_bits = 8 * sum(len(b) for fb in frames_bytes for b in fb)
    return {'frames': frames_bytes, 'model_bin': comp['final_bytes'], 'side_info': side_info, 'low_enc_bytes': low,
            'point_num': points, 'bits_est': bits_est,
            # the five one-byte side-info fields this codec adds (arith_version, precision, scale_num, block_layers,
            # hidden_channel_conv: EXTRA_SIDE_BITS = 40) are counted with the model
            'bpp': {'point_bpp': occ_bits / points, 'model_bpp': (comp['bit_real'] + EXTRA_SIDE_BITS) / points,
                    'xyzlow_bpp': len(low) * 8 / points,
                    'bpp_all': (occ_bits + comp['bit_real'] + EXTRA_SIDE_BITS + len(low) * 8) / points}}       # test_utils.py:146-157


def decode_one_frame(model, frame_enc_bytes, xyz_low):
    """decoder.decode_one_frame (decoder.py:153-176): coarse to fine, 8 AC-decoded stages per scale.  A scale is ONE call into the
    library (model.decode_scale -> linr_decode_scale: kernel map, the 7-neighbour occupancy read off it instead of
    qscTensor.set_offset_tensor's searches, the 8 stage forwards with their host round trips, upper_layer), so the Python side of a
    frame is seven buffer allocations."""
    if getattr(model, '_wide', None) is not None:          # hidden_channel_conv 16 / 32: no single-call decoder scale
        return decode_one_frame_stagewise(model, frame_enc_bytes, xyz_low)
    lowx = unique_sorted(xyz_low)
    bits = max(1, int(xyz_low.max()).bit_length()) if xyz_low.numel() else 1       # coordinates of the coarsest level
    for s_idx in range(len(frame_enc_bytes) - 1, -1, -1):
        bits = min(bits + 1, 21)
        lowx = model.decode_scale(lowx, s_idx, frame_enc_bytes[s_idx], bits)
    return {'dec_coord': lowx}


def decode_one_frame_stagewise(model, frame_enc_bytes, xyz_low):
    """The same through the reference-shaped surface (model.decode per scale + octree_level.upper_layer in torch): kept as the
    cross-check of decode_one_frame."""
    lowx = unique_sorted(xyz_low)
    for s_idx in range(len(frame_enc_bytes) - 1, -1, -1):
        occ_lst = model.decode({'enc_bytes': frame_enc_bytes[s_idx], 'coord': lowx, 'offset_tensor': None,
                                'scale_idx': s_idx})
        lowx = octree_level_obj.upper_layer(lowx, torch.cat(occ_lst, dim=-1))
    return {'dec_coord': lowx}


def decode_gop(model_ori, enc, device='cuda', frames=None, workers=1, timing=None):
    """decoder.decode_one_gop: rebuild the model from model.bin, then every frame from its streams alone.
    `timing` (a dict) receives 'setup_s': the once-per-GOP part (model.bin -> parameters, the coarsest coordinates).
    Frames are independent once the model is known; with workers > 1 they are decoded by a host thread pool, each
    thread on its own HIP stream (a frame's own chain - 56 stage forwards with a serial range decode in between - cannot
    be parallelised, but the range decoding of one frame overlaps the stage forwards and copies of the others; the
    C calls release the GIL)."""
    import time
    t_setup = time.time()
    side = dict(enc['side_info'])
    coded_with = int(side.pop('arith_version', 1))
    if coded_with != ARITH_VERSION:
        from ._lib import LinrError
        raise LinrError('this stream was coded with network arithmetic version %d; this build decodes version %d (the fp32 '
                        'accumulation order of the convolutions is part of the stream format: decode with the build that '
                        'encoded it)' % (coded_with, ARITH_VERSION))
    want = {k: side[k] for k in ('scale_num', 'block_layers', 'hidden_channel_conv') if k in side}
    have = model_shape(model_ori)
    if any(int(v) != have[k] for k, v in want.items()):
        raise ValueError('the stream was coded with a model of shape %s; the model handed to the decoder has %s' % (want, have))
    side['final_bytes'] = enc['model_bin']
    model, _ = Model_Estimate().decompress_model(model_ori, side)
    model.inference_precision = side.get('precision', 'f32')
    lows, mins = dec_all_frame_low_xyz(enc['low_enc_bytes'])
    todo = list(range(len(enc['frames'])) if frames is None else frames)
    if timing is not None:
        if device != 'cpu':
            torch.cuda.synchronize()
        timing['setup_s'] = time.time() - t_setup

    def one(i):
        xyz_low = torch.tensor(lows[i].astype(np.int32), device=device)
        dec = decode_one_frame(model, list(enc['frames'][i]), xyz_low)['dec_coord']
        return dec + torch.tensor(mins[i], device=device, dtype=torch.int32)

    if getattr(model, '_wide', None) is not None:
        workers = 1          # the channel-blocked executor keeps per-call state on the model: frames one after the other
    if workers <= 1 or len(todo) <= 1:
        return [one(i) for i in todo]
    from concurrent.futures import ThreadPoolExecutor
    main_stream = torch.cuda.current_stream()
    dev_index = torch.cuda.current_device()

    def job(i):
        torch.cuda.set_device(dev_index)
        st = torch.cuda.Stream()
        st.wait_stream(main_stream)                 # the decompressed parameters were written on the caller's stream
        with torch.cuda.stream(st):
            out = one(i)
        st.synchronize()
        return out

    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(job, todo))


def write_gop(enc, result_dir):
    bins = os.path.join(result_dir, 'bins')
    os.makedirs(bins, exist_ok=True)
    for fi, scales in enumerate(enc['frames']):
        for si, b in enumerate(scales):
            with open(os.path.join(bins, 'frame%s_scale%d.bin' % (str(fi).zfill(4), si)), 'wb') as f:
                f.write(b)
    with open(os.path.join(bins, 'model.bin'), 'wb') as f:
        f.write(enc['model_bin'])
    with open(os.path.join(bins, 'low_enc_bytes.bin'), 'wb') as f:
        f.write(enc['low_enc_bytes'])
    with open(os.path.join(result_dir, 'side_info.json'), 'w') as f:
        json.dump(enc['side_info'], f, indent=4)


def read_gop(result_dir):
    bins = os.path.join(result_dir, 'bins')
    frames, fi = [], 0
    while True:
        files = glob.glob(os.path.join(bins, 'frame%s_scale*.bin' % str(fi).zfill(4)))
        if not files:
            break
        scales = []
        for si in range(len(files)):
            with open(os.path.join(bins, 'frame%s_scale%d.bin' % (str(fi).zfill(4), si)), 'rb') as f:
                scales.append(f.read())
        frames.append(scales)
        fi += 1
    with open(os.path.join(bins, 'model.bin'), 'rb') as f:
        model_bin = f.read()
    with open(os.path.join(bins, 'low_enc_bytes.bin'), 'rb') as f:
        low = f.read()
    with open(os.path.join(result_dir, 'side_info.json')) as f:
        side = json.load(f)
    return {'frames': frames, 'model_bin': model_bin, 'low_enc_bytes': low, 'side_info': side}
