"""The bf16 training legs of bench.py (BASELINE config[4] "bf16 SparseConv"): the headline GOP through the bf16 executor, one owlii11
frame, one owlii11 GOP of 64."""
import ctypes   # noqa: F401
import json     # noqa: F401
import os
import sys      # noqa: F401
import time     # noqa: F401

import numpy as np   # noqa: F401
import torch

from .common import _time_launches
from .roofline import load_traffic, traffic_source

def bf16_train_leg(gop, L, _lib, epochs):
    """BASELINE config[4]'s "bf16 SparseConv" on the overfit: the SAME GOP trained by the bf16 training executor (linr_net_train_step_bf16:
    bf16 feature / gradient rows, fp32 master weights and accumulation) - a complete overfit from seed 8807, coded by the bf16 /
    uint8-weight codec, frames 0..1 decoded - beside the fp32 headline, never instead of it; then one frame of config[4]'s own geometry
    (owlii11: 11-bit, ~1.24 M rows) for ms/step of both executors.  `roofline` prices the executor's dominant kernel class, the fused
    backward of the convolutions 8->8 (bbwd_k<0>: backward-data + weight gradient from one gather = two algorithmic row passes of
    2 (8 + 8) + 108 bytes per group), from launch durations measured live with event pairs on the launch stream."""
    import ctypes
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    out = {'dtype': 'bf16 feature and gradient rows, fp32 master parameters / accumulation / Adam (v_mfma_f32_4x4x4_16b_bf16)'}
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    model.train_precision = 'bf16'
    gop.share_train_bf16_arena()
    opt = FlatAdam(model)
    init = model.flat_parameters().detach().clone()
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    t_ramp, i = time.time(), 0
    # clock ramp on the kernels that are about to be timed (the legs before this one are host-bound)
    while time.time() - t_ramp < 1.5:
        for _ in range(64):
            train_step(model, opt, gop.frames[i % len(gop)], gop.point_nums[i % len(gop)], out=bits)
            i += 1
        torch.cuda.synchronize()
    steps = epochs * len(gop)
    ms, runs = None, []
    # three complete overfits from the same seed (bit-identical trajectories): ALL are reported, the median is the figure
    for _ in range(3):
        model.flat_parameters().copy_(init)
        opt.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        info = {}
        e0.record()
        losses = overfit.overfit_gop(model, opt, gop, epochs,
            info=info)      # the complete overfit, un-instrumented: ms_per_step, bits/point
        e1.record()
        torch.cuda.synchronize()
        runs.append(round(e0.elapsed_time(e1) / steps, 4))
    ms = sorted(runs)[len(runs) // 2]
    # launch durations of the dominant kernel class, live (event pairs on the launch stream), from 64 more steps of a scratch copy of the
    # trained state - outside the timed overfit, whose model is what gets coded below
    snap = (model.flat_parameters().detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t, opt.t_scale.copy(), opt.lr,
        opt.sched_steps)
    L.linr_prof_mask(1 << 17)
    L.linr_prof_enable(1)
    for i in range(64):
        train_step(model, opt, gop.frames[i % len(gop)], gop.point_nums[i % len(gop)], out=bits)
    torch.cuda.synchronize()
    L.linr_prof_enable(0)
    tot, nl, npass = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
    _lib.check(L.linr_prof_read(17, ctypes.byref(tot), ctypes.byref(nl), ctypes.byref(npass)), 'linr_prof_read')
    L.linr_prof_mask(3)
    model.flat_parameters().copy_(snap[0])
    opt.exp_avg.copy_(snap[1])
    opt.exp_avg_sq.copy_(snap[2])
    opt.t, opt.t_scale, opt.lr, opt.sched_steps = snap[3], snap[4], snap[5], snap[6]
    mean_rows = sum(f.rows for f in gop.frames) / float(len(gop))
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8, precision='bf16')
    nd = min(2, len(gop))
    dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', frames=list(range(nd)), workers=nd)
    ok = all(bool(torch.equal(dec[i], torch.as_tensor(gop.infos[i]['ori']).cuda() + torch.tensor(gop.coord_mins[i], device='cuda',
        dtype=torch.int32)))
             for i in range(nd))
    out.update({'ms_per_step': round(ms, 4), 'steps': steps, 'ms_per_step_runs': runs,
                'note': 'the complete %d-epoch overfit incl. its per-epoch host reads of the loss (HIP events); the MEDIAN of the three '
                        'runs listed in ms_per_step_runs (same seed)' % epochs,
                'epoch_loss_bpp': [round(x, 4) for x in losses], 'coded_epoch': info.get('coded_epoch'),
                'bits_per_point': round(float(enc['bpp']['bpp_all']), 5), 'codec': 'bf16 features / uint8 weight codes',
                    'lossless_decode_frames0to1': ok})
    if nl.value:
        alg_row_pass = 2 * (8 + 8) + 108
        us_launch = tot.value * 1e3 / nl.value
        groups = npass.value / float(nl.value)
        achieved = groups * mean_rows * 2 * alg_row_pass / (us_launch * 1e-6) / 1e9
        # counter bytes (profiles/traffic_bf16.json): the 8-group launches bbwd_k<0,0> (prune) / <0,3> (tail), scaled to this launch mix
        tb = load_traffic('traffic_bf16.json')
        tr_launch, tr_note = None, None
        per_group = [v['bytes_per_dispatch'] / 8.0 * (mean_rows / float(tb['rows'])) for k, v in tb.get('kernels', {}).items()
                     if k.replace(' ', '').startswith('voidbbwd_k<0,3') and tb.get('rows')]
        if per_group:
            tr_launch = int(groups * per_group[0])
            tr_note = 'HBM bytes per 8-group launch of bbwd_k<0,3> / 8 x the mean groups per launch; ' + traffic_source(tb,
                'traffic_bf16.json')
        out['roofline'] = {'kernel': 'bbwd_k<0>: fused backward-data + weight gradient of the convolutions 8->8 (17 of a step\'s 33 '
                                     'backward row passes, 3 launches)',
                           'bound': 'hbm', 'peak': 8000.0, 'unit': 'GB/s',
                           'frac_counter': None if tr_launch is None else round(tr_launch / (us_launch * 1e-6) / 1e9 / 8000.0, 4),
                           'achieved_bookkeeping': round(achieved, 1), 'frac_bookkeeping': round(achieved / 8000.0, 4),
                           'alg_bytes_per_row_pass': alg_row_pass, 'row_passes_per_fused_group': 2,
                               'mean_groups_per_launch': round(groups, 3),
                           'mean_launch_us': round(us_launch, 2), 'us_per_group_pass': round(tot.value * 1e3 / max(npass.value, 1), 2),
                           'launches_sampled': int(nl.value), 'traffic': tr_launch, 'traffic_note': tr_note,
                           'frac_note': 'frac_counter (the bytes the memory system moved / launch time / 8 TB/s) is THE roofline figure '
                                        'of this executor.  '
                                        '*_bookkeeping price SURVEY 8(d)\'s algorithmic bytes: each of the two row passes a fused launch '
                                        'replaces with a 108-byte '
                                        'neighbour table, which the kernel streams as 40 bytes and once - that figure can exceed 1 and is '
                                        'no efficiency claim',
                           'step': {'alg_bytes_per_row': 20514, 'achieved_bookkeeping': round(20514 * mean_rows / (ms * 1e-3) / 1e9, 1),
                                    'frac_bookkeeping': round(20514 * mean_rows / (ms * 1e-3) / 1e9 / 8000.0, 4),
                                    'note': 'SURVEY 8(d) at 2-byte features: 3 passes x (1,654 B features + 5,184 B neighbour table) per '
                                            'row (bookkeeping, see frac_note)'}}
    del enc, dec
    # config[4]'s own geometry: one frame of the Owlii stand-in, both executors
    try:
        g4 = overfit.Gop(None, [synthetic.sequence_frame_device('owlii11', 0, 'cuda')], None, 64, 'cuda')
        res = {'rows': g4.frames[0].rows, 'points': g4.point_nums[0], 'scales': g4.scale_num}
        for prec in ('f32', 'bf16'):
            m4 = overfit.gen_model(g4.scale_num, 'cuda', seed=8807)
            m4.train_precision = prec
            o4 = FlatAdam(m4)
            for _ in range(60):
                train_step(m4, o4, g4.frames[0], g4.point_nums[0], out=bits)
            res['ms_per_step_' + prec] = round(_time_launches(lambda: train_step(m4, o4, g4.frames[0], g4.point_nums[0], out=bits),
                30) * 1e3, 4)
            del m4, o4
        res['bf16_over_f32'] = round(res['ms_per_step_bf16'] / res['ms_per_step_f32'], 3)
        out['config4_owlii11_frame'] = res
        del g4
    except Exception as e:
        out['config4_owlii11_frame'] = {'error': repr(e)}
    torch.cuda.empty_cache()
    # ... and config[4] as BASELINE states it: ONE GOP of 64 such frames, bf16 SparseConv for the overfit, the uint8 weight pack + bf16
    # features for the codec: encode sec/frame and bits/point of the whole GOP on this GPU (the config's 8 GPUs run 8 such GOPs)
    if not os.environ.get('LINR_SKIP_CONFIG4'):
        try:
            t0 = time.time()
            g64 = overfit.Gop(None, [synthetic.sequence_frame_device('owlii11', t, 'cuda') for t in range(64)], None, 64, 'cuda')
            torch.cuda.synchronize()
            stage_s = time.time() - t0
            per = {}
            # the config's precision first; the fp32 executor from the SAME seed beside it
            for prec in ('bf16', 'f32'):
                m64 = overfit.gen_model(g64.scale_num, 'cuda', seed=8807)
                m64.train_precision = prec
                o64 = FlatAdam(m64)
                info64 = {}
                torch.cuda.synchronize()
                t0 = time.time()
                l64 = overfit.overfit_gop(m64, o64, g64, epochs, info=info64)
                torch.cuda.synchronize()
                t1 = time.time()
                e64 = codec.encode_gop(m64, overfit.gen_model(g64.scale_num, 'cuda'), g64, 8, precision='bf16')
                torch.cuda.synchronize()
                t2 = time.time()
                d64 = codec.decode_gop(overfit.gen_model(g64.scale_num, 'cuda'), e64, 'cuda', frames=[0, 63], workers=2)
                ok64 = all(bool(torch.equal(d, torch.as_tensor(g64.infos[i]['ori']).cuda() + torch.tensor(g64.coord_mins[i],
                    device='cuda', dtype=torch.int32)))
                           for d, i in zip(d64, (0, 63)))
                if not ok64:
                    raise RuntimeError('config4_gop64: the %s-trained GOP did not decode losslessly' % prec)
                per[prec] = {'encode_sec_per_frame': round((t2 - t0) / 64.0, 5), 'overfit_s': round(t1 - t0, 3), 'codec_s': round(t2 - t1,
                    3),
                             'ms_per_step': round((t1 - t0) * 1e3 / (epochs * 64), 4),
                                 'bits_per_point': round(float(e64['bpp']['bpp_all']), 5),
                             'epoch_loss_bpp': [round(x, 4) for x in l64], 'coded_epoch': info64.get('coded_epoch'),
                                 'lossless_decode_frames_0_63': ok64}
                del m64, o64, e64, d64
            out['config4_gop64'] = {'workload': 'BASELINE config[4] stand-in: synthetic owlii11 (11-bit sphere shell, %d points and %d '
                                                'rows in frame 0, %d scales), ONE GOP of 64 '
                                                'frames, %d epochs, bf16 / uint8-weight codec; bf16 training (the config) and fp32 '
                                                'training from the same seed 8807'
                                                % (g64.point_nums[0], g64.frames[0].rows, g64.scale_num, epochs),
                                    'staging_s': round(stage_s, 2), 'bf16_training': per['bf16'], 'f32_training': per['f32'],
                                    'encode_sec_per_frame': per['bf16']['encode_sec_per_frame'],
                                        'bits_per_point': per['bf16']['bits_per_point'],
                                    'bits_per_point_bf16_over_f32': round(per['bf16']['bits_per_point'] / per['f32']['bits_per_point'], 4),
                                    'lossless_both_precisions': True,
                                    'note': 'encode = overfit + codec of the whole GOP on this one GPU, inputs resident (the bf16 run '
                                            'includes the first codec call of this GOP '
                                            'size, which sizes the pinned staging ring); ONE seed: one overfit is chaotic in the '
                                            'rounding, the ratio is a sample, not a bound'}
            print('config4_gop64: bf16 training %.5f bpp, fp32 training %.5f bpp (seed 8807), both lossless'
                  % (per['bf16']['bits_per_point'], per['f32']['bits_per_point']),
                  file=sys.stderr, flush=True)
            del g64
        except Exception as e:
            out['config4_gop64'] = {'error': repr(e)}
    torch.cuda.empty_cache()
    return out


def config4_rank_leg(rank, world, dist, barrier, epochs, frames=64):
    """BASELINE config[4] as stated - owlii11 GOPs of 64 frames, bf16 SparseConv + uint8 weight pack, one GOP per GPU: EVERY rank stages,
    trains (bf16 executor) and codes (bf16 / uint8-weight codec) ITS OWN GOP (frames rank * 64 ...), no collective on the data path;
    barrier + synchronize on both sides of the timed region, MAX over ranks.  value = 64 x world frames / that time (weak scaling).
    Only with world > 1: on one GPU bf16_train.config4_gop64 is this measurement."""
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.gop_parallel import max_over_ranks, sum_over_ranks
    from linr_pcgc_amd.model_core import FlatAdam
    first = rank * frames
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('owlii11', first + t, 'cuda') for t in range(frames)], None, 64, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    model.train_precision = 'bf16'
    opt = FlatAdam(model)
    overfit.overfit_gop(model, opt, gop.subset(2), 1)                      # first-call costs of this GOP size, then back to the seed
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    model.train_precision = 'bf16'
    opt = FlatAdam(model)
    barrier()
    t0 = time.time()
    overfit.overfit_gop(model, opt, gop, epochs)
    torch.cuda.synchronize()
    t1 = time.time()
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8, precision='bf16')
    torch.cuda.synchronize()
    t2 = time.time()
    barrier()
    wall = max_over_ranks(time.time() - t0, dist, 'cuda')
    dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', frames=[0])
    ok = bool(torch.equal(dec[0], torch.as_tensor(gop.infos[0]['ori']).cuda() + torch.tensor(gop.coord_mins[0], device='cuda',
        dtype=torch.int32)))
    all_ok = sum_over_ranks(1.0 if ok else 0.0, dist, 'cuda') == world
    bits = sum_over_ranks(float(enc['bpp']['bpp_all']) * sum(gop.point_nums), dist, 'cuda')
    pts = sum_over_ranks(float(sum(gop.point_nums)), dist, 'cuda')
    overfit_s = max_over_ranks(t1 - t0, dist, 'cuda')
    codec_s = max_over_ranks(t2 - t1, dist, 'cuda')
    del gop, enc, dec
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {'workload': 'BASELINE config[4] stand-in: synthetic owlii11, ONE GOP of %d frames PER RANK (rank r: frames %d r ...), %d '
                        'epochs of '
                        'bf16 training, bf16 / uint8-weight codec, seed 8807 on every rank' % (frames, frames, epochs),
            'n_gpus': world, 'scaling': 'weak', 'wall_s_max_over_ranks': round(wall, 4),
            'encode_sec_per_frame': round(wall / (frames * world), 6), 'frames_per_s_whole_job': round(frames * world / wall, 2),
            'overfit_s_max': round(overfit_s, 3), 'codec_s_max': round(codec_s, 3), 'bits_per_point_all_ranks': round(bits / pts, 5),
            'lossless_decode_frame0_every_rank': bool(all_ok)}
