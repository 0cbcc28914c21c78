#!/usr/bin/env python3
"""Bench: BASELINE.json's metric (encode sec/frame + bits/point, lossless) on its config[1] stand-in:
synthetic "loot10" (10-bit sphere, ~784 k points/frame, 336 k parent rows over 7 scales), 1 GOP of 32 frames,
first_epoch = 10, one GOP per GPU (no data-path collective).

A step = one frame-epoch of the per-GOP overfit (forward + backward + fused Adam + StepLR on one frame), the unit the
reference logs as train_time_avg (loot/info.log).  The K timed steps walk the GOP's frames in the reference's order
(main.py:297-321); `ms_per_step` is their mean.  Whatever K is, the overfit is then carried on to its full
epochs x frames steps (timed as a second region), so `bits_per_point` and `value` always describe the SAME complete
training:
    value = (full overfit wall / frames) + codec_time_per_frame     [s/frame, whole job: divided by the number of GPUs]
The codec leg (model compression, one inference forward + D2H + range coding + stream files per frame) is timed
separately, frames are decoded and checked bit-exact.  Warm-up runs the exact timed-loop body (live kernel timing and
loss accumulation included) and the state is reset in place, so nothing idles the GPU between warm-up and t0; every
timed step also gets a HIP event so the log shows the per-step spread.

After the headline the BASELINE config[2] flow (300 frames, GOP 32, GOP 0 as serial prefix, GOPs sharded over the GPUs
with no collective) runs once for real and is reported as `sequence` (whole-sequence wall, phase-B efficiency, ideal
bound); `--sequence` makes that run the headline (strong scaling).  On one GPU the full-size oracle gradient that the
CPU baseline computes anyway is compared with the HIP forward/backward tensor by tensor.

    python bench.py --gpus 1 --steps 320 --warmup 32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# noqa: E402
from bench.bf16_train import bf16_train_leg, config4_rank_leg
from bench.common import EPOCHS, TABLE_STEPS, host_threads, log, parse                                            # noqa: E402
from bench.cpu_oracle import cpu_baseline, full_size_parity                                                       # noqa: E402
from bench.headline import Headline                                                                               # noqa: E402
from bench.legs import bf16_codec_leg, bpp_seeds_leg, codec_leg, decode_leg, device_report, sequence_leg, wide_leg   # noqa: E402
from bench.roofline import kernel_roofline                                                                        # noqa: E402
from bench.rough import rough_leg                                                                                 # noqa: E402


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks the way the driver does (child process; nothing has touched
        # the GPU yet) and pass its exit code on
        import subprocess
        port = os.environ.get('MASTER_PORT', '29533')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('LINR_BENCH_SINGLE_DEVICE'):          # rehearsal of the N > 1 control flow on a 1-GPU box
        local = 0
    assert torch.cuda.is_available(), 'bench.py needs an MI355X: the coding network has no CPU path'
    torch.set_num_threads(host_threads())          # torch CPU ops otherwise fan out over every core of the node, per rank
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        from linr_pcgc_amd.run import init_dist
        dist = init_dist(local)
    from linr_pcgc_amd import _lib, synthetic
    L = _lib.lib()
    devices = device_report(world, dist, local)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.sequence:
        # headline = the whole sequence (strong scaling); a short ramp so the first GOP does not start at idle clocks
        seq = sequence_leg(args, rank, world, dist)
        if rank == 0:
            out = {'metric': 'encode_sec_per_frame', 'value': seq['sec_per_frame'], 'unit': 's/frame', 'n_gpus': world,
                   'steps': args.seq_frames * args.seq_epochs, 'warmup': 0,
                   'ms_per_step': round(seq['wall_s'] * 1e3 / (args.seq_frames * args.seq_epochs), 4),
                   'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                   'config': {'workload': seq['workload'], 'parallelism': 'gop-per-gpu x%d (no collective)' % world},
                   'bits_per_point': seq['bits_per_point'], 'sequence': seq, 'roofline': None, 'cpu_baseline': None, 'devices': devices}
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- headline: K timed steps of the overfit, then the rest of the complete overfit -------------------------------------------------
    h = Headline(args, rank, L, _lib)
    h.run(barrier, dist)
    gop = h.gop
    table_prof = h.kernel_table_leg() if rank == 0 and not os.environ.get('LINR_SKIP_ROOFLINE') else None
    # ---- codec, decoder, the legs reported beside the headline ------------------------------------------------------------------------
    enc, codec_s, codec_cold_s = codec_leg(h, rank, dist, barrier)
    codec_s_per_frame = codec_s / len(gop)
    lossless, decode_s, decode_pts, nd = decode_leg(h, enc)
    wide = wide_leg(h) if rank == 0 and not os.environ.get('LINR_SKIP_WIDE') else None
    bf16_train = None
    if rank == 0 and not os.environ.get('LINR_SKIP_BF16_TRAIN'):
        try:
            bf16_train = bf16_train_leg(gop, L, _lib, EPOCHS)
        except Exception as e:
            bf16_train = {'error': repr(e)}
        log('bf16 training leg: %s' % bf16_train)
    bf16_leg, bf_lossless = bf16_codec_leg(h, enc, nd, barrier)
    lossless = lossless and bf_lossless
    bpp_seeds = bpp_seeds_leg(h, enc) if rank == 0 and not os.environ.get('LINR_SKIP_BPP_SEEDS') else None
    rough = None
    if rank == 0 and not os.environ.get('LINR_SKIP_ROUGH'):
        try:
            mean_rows = sum(f.rows for f in gop.frames) / float(len(gop))
            bf_ms = bf16_train.get('ms_per_step') if isinstance(bf16_train, dict) else None
            rough = rough_leg(EPOCHS, min(len(gop), 32), mean_rows, h.ms_per_step, bf_ms)
        except Exception as e:
            rough = {'error': repr(e)}
            log('rough leg failed: %r' % (e,))

    overfit_s_per_frame = h.full_overfit_s / len(gop)
    value = (overfit_s_per_frame + codec_s_per_frame) / world
    out = None
    if rank == 0:
        out = {'metric': 'encode_sec_per_frame', 'value': round(value, 5), 'unit': 's/frame', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(h.ms_per_step, 4),
               'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE config[%s] stand-in: synthetic %s (%s, %d points and %d parent rows in frame 0, %d '
                                      'scales), '
                                      '1 GOP of %d frames per GPU, first_epoch=%d, lr 0.01 StepLR(32,0.992) Adam wd 1e-4, seed 8807'
                                      % ({'sphere8': '0', 'loot10': '1', 'andrew10': '3', 'owlii11': '4'}.get(args.config, '?'),
                                          args.config,
                                         synthetic.describe(args.config), gop.point_nums[0], gop.frames[0].rows, gop.scale_num, len(gop),
                                             EPOCHS),
                          'frames_per_gpu': len(gop), 'epochs': EPOCHS, 'parallelism': 'gop-per-gpu x%d (no collective)' % world},
               'devices': devices,
               'value_note': 'overfit (complete %d epochs) + the steady-state codec call, per frame; a process\'s FIRST codec call also '
                             'pays for pinned buffers and coder threads: value_cold below uses it' % EPOCHS,
               'value_cold': round((overfit_s_per_frame + codec_cold_s / len(gop)) / world, 5),
               'bits_per_point': round(enc['bpp']['bpp_all'], 5),
               'bits_per_point_seeds': bpp_seeds,
               'bits_per_point_after_steps': h.steps_done,
               'coded_epoch': h.coded_epoch, 'coded_epoch_policy': 'best mean loss of the overfit (main.py:413-426); epochs count from 0',
               'bpp_components': {k: round(v, 6) for k, v in enc['bpp'].items()},
               'lossless_decode_frames0to3': lossless,
               'full_overfit': {'steps': h.steps_done, 'seconds': round(h.elapsed + h.rest_s, 4),
                                'ms_per_step': round((h.elapsed + h.rest_s) * 1e3 / h.steps_done, 4),
                                'note': 'value and bits_per_point both come from this complete %d-epoch overfit; ms_per_step is '
                                        'the mean of its first `steps` steps' % EPOCHS},
               'per_step_ms_hip_events': h.step_stats,
               'components_s_per_frame': {'overfit': round(overfit_s_per_frame, 5),
                   'codec_modelcomp_fwd_ac_write': round(codec_s_per_frame, 5),
                                          'codec_first_call': round(codec_cold_s / len(gop), 5),
                                          'decode_s_per_frame_4_in_flight': round(decode_s, 4),
                                          'decode_s_single_frame': round(decode_pts.get(1, 0.0), 4),
                                          'decode_s_per_frame_8_in_flight': round(decode_pts.get(8, 0.0), 4),
                                          'decode_gop_setup_s': round(decode_pts.get('gop_setup_s', 0.0), 4)},
               'bf16_codec': bf16_leg,
               'bf16_train': bf16_train,
               'rough': rough,
               'hidden16': wide,
               'epoch_loss_bpp': [round(x, 4) for x in h.losses], 'setup_s': round(h.setup_s, 1),
               'reference_logged': {'train_s_per_frame_epoch': 0.55, 'codec_s_per_frame': 0.43,
                                    'source': 'loot/info.log, loot/gop_32_62/*/result.json (RTX 3090, real loot)'}}
        out['roofline'] = None if os.environ.get('LINR_SKIP_ROOFLINE') else kernel_roofline(gop, h.live, table_prof, TABLE_STEPS,
            h.ms_per_step)
        log('roofline: %s' % out['roofline'])
    config4_ranks = None
    # every rank its own owlii11 GOP of 64 (bf16): configs [2] and [4] in one N-rank record
    if world > 1 and not os.environ.get('LINR_SKIP_CONFIG4'):
        try:
            config4_ranks = config4_rank_leg(rank, world, dist, barrier, EPOCHS, frames=int(os.environ.get('LINR_CONFIG4_FRAMES', 64)))
            log('config[4] per rank: %s' % config4_ranks)
        except Exception as e:
            config4_ranks = {'error': repr(e)}
            print('[bench rank %d] config[4] per-rank leg failed: %r' % (rank, e), file=sys.stderr, flush=True)
    if rank == 0:
        out['config4_per_rank'] = config4_ranks
    parity_ok = True
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            torch.set_num_threads(host_threads())
            log('cpu baseline on %d threads ...' % host_threads())
            out['cpu_baseline'], o_bits, o_grads = cpu_baseline(h.init_sd, gop.infos[0], gop.point_nums[0], args.cpu_sample_rows)
            out['full_size_parity'] = full_size_parity(h.init_sd, gop.frames[0], gop.point_nums[0], o_bits, o_grads, gop.scale_num)
            parity_ok = out['full_size_parity']['ok']
            log('full-size parity vs oracle: %s' % out['full_size_parity'])
        else:
            out['cpu_baseline'] = None
    # free the headline's GOP before the sequence leg stages its own frames
    del gop, enc, h
    torch.cuda.empty_cache()
    seq = None
    if not args.no_sequence:
        try:
            seq = sequence_leg(args, rank, world, dist)
            log('sequence leg: %s' % seq)
        except Exception as e:          # the headline above is already measured: report the failure instead of losing the line
            seq = {'error': repr(e)}
            log('sequence leg failed: %r' % (e,))
            if rank != 0:               # log() speaks for rank 0 only: a failure on another rank must still be visible
                import traceback
                print('[bench rank %d] sequence leg failed:\n%s' % (rank, traceback.format_exc()), file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # the same flow END TO END: nothing resident before t0 - every GOP is generated, its octrees and kernel maps built (the next GOP's in
    # a background thread while the current one trains), overfitted, encoded, spot-decoded.  One GPU only (with N > 1 the headline
    # numbers above already came from N ranks; the cold flow is a per-rank property).
    seq_cold, stage_split = None, None
    if world == 1 and not args.no_sequence and not os.environ.get('LINR_SKIP_COLD'):
        try:
            seq_cold = sequence_leg(args, rank, world, dist, stage_all=False)
            log('cold sequence leg: %s' % seq_cold)
            from linr_pcgc_amd import overfit as _ov
            _ov.staging_split(args.config, range(2), 'cuda')                    # first calls pay for lazy loads
            stage_split = _ov.staging_split(args.config, range(8), 'cuda')
            stage_split['note'] = ('ms per frame, each step timed synchronously on frames 0..7: the synthetic generator stands in for file '
                                   'input; octree = minimum / sort + unique / child occupancy of every level (linr_coords_minmax, '
                                   'linr_coords_sort_unique, linr_octree_levels); kernel_map = neighbour search, compressed map, tiled '
                                   'copy, '
                                   '7-neighbour features')
            log('staging split: %s' % stage_split)
        except Exception as e:
            seq_cold = {'error': repr(e)}
            log('cold sequence leg failed: %r' % (e,))
    if rank == 0:
        out['sequence'] = seq
        out['sequence_cold'] = seq_cold
        out['staging_ms_per_frame'] = stage_split
        if isinstance(seq_cold, dict) and 'sec_per_frame' in seq_cold:
            out['sequence_cold_sec_per_frame'] = seq_cold['sec_per_frame']
        if isinstance(seq, dict) and 'sec_per_frame' in seq:
            # strong-scaling numbers of the BASELINE config[2] flow at the TOP level, beside the weak-scaling `value`
            out['sequence_sec_per_frame'] = seq['sec_per_frame']
            out['sequence_wall_s'] = seq['wall_s']
            out['phase_b_efficiency'] = seq['phase_b_efficiency']
            out['ideal_speedup_bound'] = seq['ideal_speedup_bound']
        assert lossless, 'decoded geometry differs from the input'
        print(json.dumps(out))
        assert parity_ok, 'HIP forward/backward differs from the CPU oracle at full size: %s' % out.get('full_size_parity')


if __name__ == '__main__':
    main()
