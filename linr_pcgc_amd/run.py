"""Sequence driver on the fast path: the overfit -> encode -> decode flow of main.overfit_enc_dec (main.py:69-119), one
process per GPU (GOP sharding of gop_parallel.py, no data-path collective).

    python -m linr_pcgc_amd.run --config loot10 --frames 300 --gop 32 --first-epoch 10 --others-epoch 10 --out /tmp/linr_out
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m linr_pcgc_amd.run ...

``run_sequence_job`` is the same flow as a function (bench.py --sequence and the tests call it).
"""
import argparse
import datetime
import json
import os
import time

import torch

from . import codec, gop_parallel, overfit, ply, synthetic
from .model_core import FlatAdam


def parse(argv=None):
    ap = argparse.ArgumentParser('linr_pcgc_amd.run')
    ap.add_argument('--config', default='loot10')
    ap.add_argument('--input-glob', default=None, help="PLY / npy frames of a real sequence (sorted by name), e.g. '/data/loot/Ply/*.ply'; replaces --config")
    ap.add_argument('--frames', '--frame_num', dest='frames', type=int, default=32)
    ap.add_argument('--gop', '--gop_size', dest='gop', type=int, default=32)
    ap.add_argument('--first-epoch', '--first_epoch', dest='first_epoch', type=int, default=10)
    ap.add_argument('--others-epoch', '--others_epoch', dest='others_epoch', type=int, default=10)
    ap.add_argument('--learning-rate', '--learning_rate', dest='learning_rate', type=float, default=0.01)
    ap.add_argument('--gamma', type=float, default=0.992)
    ap.add_argument('--step-size', '--step_size', dest='step_size', type=int, default=32)
    ap.add_argument('--min-lr', '--min_lr', dest='min_lr', type=float, default=4e-4)
    ap.add_argument('--decay-rate', '--decay_rate', dest='decay_rate', type=float, default=1e-4)
    ap.add_argument('--block-layers', '--block_layers', dest='block_layers', type=int, default=1)
    ap.add_argument('--hidden-channel-conv', '--hidden_channel_conv', dest='hidden_channel_conv', type=int, default=8, choices=[8, 16, 32],
                    help='main.py:520; 8 = the tuned kernels, 16 / 32 = the channel-blocked executor (several times slower)')
    ap.add_argument('--seed', type=int, default=8807)
    ap.add_argument('--out', '--result_dir', dest='out', default='/tmp/linr_out')
    ap.add_argument('--ori_dir', '--ori-dir', dest='ori_dir', default=None, help='main.py --ori_dir: a directory of frames (with --ori_dtype), sorted by name; same as --input-glob DIR/*.TYPE')
    ap.add_argument('--ori_dtype', '--ori-dtype', dest='ori_dtype', default='ply', choices=['ply', 'npy'])
    ap.add_argument('--min_point_num', '--min-point-num', dest='min_point_num', type=int, default=64, help='main.py --min_point_num: the octree stops below this many voxels')
    ap.add_argument('--scale_num', '--scale-num', dest='scale_num', type=int, default=None, help='main.py --scale_num: at most this many scales (default: down to --min_point_num, fixed by frame 0 of a GOP)')
    ap.add_argument('--model_bitdepth', '--model-bitdepth', dest='model_bitdepth', type=int, default=8, help='main.py --model_bitdepth: bits of the weight quantiser')
    ap.add_argument('--schedule', default='pull', choices=['pull', 'static'])
    ap.add_argument('--no-stage-ahead', dest='stage_ahead', action='store_false',
                    help='stage a GOP (file parsing, octrees, kernel maps) only when it is about to run instead of in the background during the GOP before it')
    ap.add_argument('--keep', default='best', choices=['best', 'last'],
                    help="which epoch's model is coded / handed to the next GOPs: the one with the lowest mean loss (the reference: main.py:413-426) or the last")
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16'],
                    help='arithmetic of the coding forward: f32, or bf16 features with the uint8 weight codes de-quantised in-kernel (BASELINE config[4]); travels in side_info.json')
    ap.add_argument('--train-precision', '--train_precision', dest='train_precision', default=None, choices=['f32', 'bf16'],
                    help='arithmetic of the overfit step: f32 (the headline), or bf16 feature / gradient rows with fp32 master weights and '
                         'accumulation (BASELINE config[4] "bf16 SparseConv"; hidden_channel_conv 8, block_layers 1).  Default: follows --precision')
    ap.add_argument('--decode', action='store_true', help='decode every GOP again and check it is lossless')
    ap.add_argument('--mid-test', action='store_true',
                    help='main.py --mid_test: measure the model through Test_one_gop (model.codec) at epochs 0..9 and every --check-freq-th '
                         'epoch of every GOP; results under <out>/output/<gop>/<epoch>/ and <out>/output/<gop>/result.json')
    ap.add_argument('--check-freq', '--check_freq', dest='check_freq', type=int, default=5, help='main.py --check_freq')
    ap.add_argument('--write-real-bitstream', action='store_true', help='main.py --write_real_bitstream: the mid-test also writes its bins at every 50th epoch')
    return ap.parse_args(argv)


def train_precision(args):
    """--train-precision, or what --precision says when it is not given (--precision bf16 = BASELINE config[4]: bf16 SparseConv for
    the overfit AND the codec); the wide models (hidden_channel_conv 16 / 32) and deeper block_in variants train in fp32 only."""
    tp = getattr(args, 'train_precision', None)
    if tp is None:
        tp = getattr(args, 'precision', 'f32')
        if getattr(args, 'hidden_channel_conv', 8) != 8 or getattr(args, 'block_layers', 1) != 1:
            tp = 'f32'
    return tp


def run_sequence_job(args, rank=0, world=1, dist=None, stage_all=False, files=None, decode_frames=None):
    """Overfit + encode (+ decode check) of a whole sequence, sharded GOP-per-GPU.

    stage_all=True: every GOP this rank will run under the static schedule is staged in HBM first and the timed
    region starts behind a barrier with all inputs resident (bench.py's contract); otherwise GOPs are staged as they are
    claimed (ranks >= 1 stage their first GOP while rank 0 runs GOP 0).  decode_frames: None = all frames of a GOP when
    args.decode, or an int = the first n frames of each GOP.
    Returns (summary dict on every rank, {gop index: result} of this rank)."""
    groups = gop_parallel.split_gops(args.frames, args.gop)
    device = 'cuda'

    # Staging (file read + parse on a thread pool, octrees, kernel maps) of this rank's NEXT GOP runs in a background thread on its own
    # HIP stream while the current GOP trains, wherever the next GOP is known in advance: one rank, or the static deal.  (With the
    # pull queue and several ranks the next GOP belongs to whoever is free first, so nothing is claimed ahead of time.)
    schedule = 'static' if stage_all else getattr(args, 'schedule', 'pull')
    ahead = None
    dev_index = torch.cuda.current_device()
    if not stage_all and getattr(args, 'stage_ahead', True):
        from concurrent.futures import ThreadPoolExecutor
        if world == 1:
            my_order = [0] + gop_parallel.phase_b_order(groups)
        elif schedule == 'static':
            my_order = ([0] if rank == 0 else []) + gop_parallel.assign_gops(groups, world)[rank]
        else:
            my_order = []
        if len(my_order) > 1:
            ahead = {'pool': ThreadPoolExecutor(max_workers=1), 'futures': {}, 'stream': torch.cuda.Stream(),
                     'next': {tuple(groups[a]): groups[b] for a, b in zip(my_order, my_order[1:])}}

    def load_group(group):
        if files is None:
            return [synthetic.sequence_frame_device(args.config, t, device) for t in group]
        return ply.read_many([files[t] for t in group])

    # main.py:73-78: the scale count of the whole sequence is fixed by its FIRST frame (dataset[0]) unless --scale_num gives it; every
    # GOP's model has that many scale embeddings / scale MLPs (GOPs >= 1 load GOP 0's checkpoint), frames that run out of voxels
    # earlier simply have fewer scales, frames that could go deeper stop there.  Every rank derives it from frame 0 itself.
    seq_scale_num = getattr(args, 'scale_num', None)
    if seq_scale_num is None:
        from .module_utils import prepare_frame
        first = ply.read_points(files[0]) if files is not None else synthetic.sequence_frame_device(args.config, 0, device)
        seq_scale_num = prepare_frame(first, None, getattr(args, 'min_point_num', 64), device=device, with_offsets=False)['scale_num']
        del first

    def build_gop(group):
        return overfit.Gop(None, load_group(group), seq_scale_num, getattr(args, 'min_point_num', 64), device,
                           block_layers=getattr(args, 'block_layers', 1))

    def build_gop_ahead(group):
        torch.cuda.set_device(dev_index)       # a new thread starts on device 0, whatever the rank's device is
        with torch.cuda.stream(ahead['stream']):
            gop = build_gop(group)
        ahead['stream'].synchronize()          # everything the GOP holds is complete before another stream reads it
        return gop

    def make_opt(model):
        return FlatAdam(model, lr=args.learning_rate, weight_decay=args.decay_rate, step_size=args.step_size, gamma=args.gamma)

    def stage(group):
        t0 = time.time()
        fut = ahead['futures'].pop(tuple(group), None) if ahead is not None else None
        if fut is not None:
            gop = fut.result()                 # staged in the background (its wait is what is left of it)
        else:
            gop = build_gop(group)
            torch.cuda.synchronize()
        if ahead is not None:
            nxt = ahead['next'].get(tuple(group))
            if nxt is not None and tuple(nxt) not in ahead['futures']:
                ahead['futures'][tuple(nxt)] = ahead['pool'].submit(build_gop_ahead, nxt)
        return gop, time.time() - t0

    def run_gop(group, epochs, ckpt, staged):
        gop, stage_s = staged if staged is not None else stage(group)
        t0 = time.time()
        model = overfit.gen_model(gop.scale_num, device, seed=args.seed, block_layers=getattr(args, 'block_layers', 1), hidden=getattr(args, 'hidden_channel_conv', 8))
        model.train_precision = train_precision(args)
        opt = make_opt(model)
        if ckpt is not None:
            overfit.warm_start(model, opt, ckpt)                 # main.py:241-248
        info = {}
        mid = []

        gop_dir = os.path.join(args.out, 'output', gop_parallel.gop_name(group))
        os.makedirs(gop_dir, exist_ok=True)
        log = open(os.path.join(args.out, 'info.log' if rank == 0 else 'info_rank%d.log' % rank), 'a')
        log.write('=' * 40 + '\nprocess_file: %d %d\n' % (group[0], group[-1]))
        clock = {'mark': time.time(), 'train': 0.0}

        def on_epoch(epoch, loss_mean):
            # main.py:327-338,428-430: the epoch's record in info.log and in <gop>/result.json (train_time: cumulative seconds of the
            # frame loops of this GOP; train_time_avg: per frame).  overfit_gop has just read the loss, so the GPU is idle here.
            clock['train'] += time.time() - clock['mark']
            entry = {'epoch': epoch, 'loss': loss_mean, 'train_time': clock['train'], 'train_time_avg': clock['train'] / len(group)}
            log.write('epoch: %d\nloss: %r\ntrain_time: %r\ntrain_time_avg: %r\n' % (epoch, loss_mean, entry['train_time'], entry['train_time_avg']))
            if getattr(args, 'mid_test', False) and (epoch < 10 or epoch % args.check_freq == 0):
                # main.py:341-411: with --mid_test the checkpoint is written at every tested epoch (epochs 0..9 and every
                # check_freq-th) and Test_one_gop measures it through model.codec into <gop>/<epoch>/ (result.json, side_info.json;
                # the bitstream files too at every 50th epoch with --write-real-bitstream)
                from .model_codec import Model_Estimate
                from .test_utils import Test_one_gop
                path = os.path.join(gop_dir, 'model_mid.pth')
                torch.save(overfit.checkpoint(model, opt, epoch, loss_mean, getattr(args, 'model_bitdepth', 8)), path)
                gen = lambda: overfit.gen_model(gop.scale_num, device, block_layers=getattr(args, 'block_layers', 1),
                                                hidden=getattr(args, 'hidden_channel_conv', 8))
                out = Test_one_gop({'model_path': path, 'Gen_Model': gen, 'frame_num': len(group),
                                    'compress_model_test': Model_Estimate().compress_test,
                                    'reading_data': [{'all_input_info': fr['all_input_info'], 'point_num': fr['point_num']} for fr in gop.infos],
                                    'result_dir': os.path.join(gop_dir, str(epoch)),
                                    'write_flag': bool(args.write_real_bitstream and epoch % 50 == 0),
                                    'low_enc_ret': codec.enc_all_frame_low_xyz(gop)})
                entry.update({'real_bpp_all': out['bpp_all'], 'real_point_bpp': out['point_bpp'], 'point_bpp_val': out['point_bpp_val'],
                              'model_bpp': out['model_bpp'], 'xyzlow_bpp': out['xyzlow_bpp'], 'enc_time': out['enc_time'],
                              'dec_time': out['dec_time'], 'enc_mode': out['enc_mode'],
                              'model_bitdepth_final': getattr(args, 'model_bitdepth', 8)})     # main.py:411 writes the depth in use
                for key in ('real_bpp_all', 'real_point_bpp', 'point_bpp_val', 'model_bpp', 'xyzlow_bpp', 'enc_time', 'dec_time', 'enc_mode'):
                    log.write('%s: %r\n' % (key, entry[key]))
            log.write('\n')
            log.flush()
            mid.append(entry)
            with open(os.path.join(gop_dir, 'result.json'), 'w') as f:
                json.dump(mid, f, indent=4)
            clock['mark'] = time.time()

        try:
            losses = overfit.overfit_gop(model, opt, gop, epochs, args.min_lr, keep=getattr(args, 'keep', 'best'), info=info, on_epoch=on_epoch)
        finally:
            log.close()
        torch.cuda.synchronize()
        t1 = time.time()
        enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, device, block_layers=getattr(args, 'block_layers', 1), hidden=getattr(args, 'hidden_channel_conv', 8)), gop,
                               getattr(args, 'model_bitdepth', 8), precision=getattr(args, 'precision', 'f32'))
        res_dir = os.path.join(args.out, 'result_enc', gop_parallel.gop_name(group))
        codec.write_gop(enc, res_dir)
        torch.cuda.synchronize()
        t2 = time.time()
        ok = None
        if args.decode:
            todo = list(range(len(group))) if decode_frames is None else list(range(min(decode_frames, len(group))))
            dec = codec.decode_gop(overfit.gen_model(gop.scale_num, device, block_layers=getattr(args, 'block_layers', 1), hidden=getattr(args, 'hidden_channel_conv', 8)),
                                   codec.read_gop(res_dir), device, frames=todo, workers=4)
            ok = all(torch.equal(d, torch.as_tensor(gop.infos[i]['ori']).cuda() +
                                 torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32))
                     for d, i in zip(dec, todo))
        torch.cuda.synchronize()
        t3 = time.time()
        result = {'gop': gop_parallel.gop_name(group), 'frames': len(group), 'epochs': epochs, 'loss': losses,
                  'coded_epoch': info['coded_epoch'], 'coded_loss': info['coded_loss'],
                  'bpp': enc['bpp'], 'points': enc['point_num'], 'lossless': ok, 'stage_s': stage_s, 'overfit_s': t1 - t0,
                  'encode_s': t2 - t1, 'decode_s': t3 - t2, 'seconds': stage_s + (t3 - t0), 'rank': rank}
        if getattr(args, 'mid_test', False):
            result['mid_test'] = mid
        del gop
        return model, opt, losses, result, info

    def first_fn(group, staged=None):
        model, opt, losses, result, info = run_gop(group, args.first_epoch, None, staged)
        ck = overfit.checkpoint(model, opt, info['coded_epoch'], info['coded_loss'], getattr(args, 'model_bitdepth', 8))      # the state overfit_gop left: the kept epoch
        ck['result'] = result
        return ck

    def other_fn(group, ckpt, staged=None):
        return run_gop(group, args.others_epoch, ckpt, staged)[3]

    prepared = {}
    t_stage0 = time.time()
    if stage_all:
        mine = ([0] if rank == 0 else []) + gop_parallel.assign_gops(groups, world)[rank]
        for g in mine:
            prepared[g] = stage(groups[g])
    stage_all_s = time.time() - t_stage0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    work_dir = os.path.join(args.out, 'output')
    try:
        try:
            results = gop_parallel.run_sequence(groups, work_dir, first_fn, other_fn, rank, world, dist,
                                                prepare_fn=stage, schedule=schedule, prepared=prepared, done_marker=False)
        finally:
            if ahead is not None:
                ahead['pool'].shutdown(wait=True)
        torch.cuda.synchronize()                 # an asynchronous GPU error of the last GOP surfaces here ...
    except BaseException:
        gop_parallel.mark_failed(work_dir, rank)         # ... and releases the other ranks from wait_all_done
        raise
    gop_parallel.mark_done(work_dir, rank)       # only now: this rank will reach the final reductions
    my_wall = time.time() - t0
    # every rank has finished (its done marker) or one has failed (its marker: raise here, outside any collective)
    gop_parallel.wait_all_done(work_dir, world)
    wall = gop_parallel.max_over_ranks(my_wall, dist, 'cuda')
    # phase A = rank 0's GOP 0; phase B = the rest of the wall.  Phase-B efficiency = busy GPU-seconds of the GOPs >= 1
    # over (ranks x phase-B wall): what SURVEY.md section 8e asks to be reported next to the whole-sequence wall.
    timed = lambda r: r['overfit_s'] + r['encode_s'] + r['decode_s'] + (0.0 if stage_all else r['stage_s'])
    phase_a = timed(results[0]) if 0 in results else 0.0
    phase_a = gop_parallel.max_over_ranks(phase_a, dist, 'cuda')
    busy_b = gop_parallel.sum_over_ranks(sum(timed(r) for g, r in results.items() if g != 0), dist, 'cuda')
    bits = gop_parallel.sum_over_ranks(sum(r['bpp']['bpp_all'] * r['points'] for r in results.values()), dist, 'cuda')
    points = gop_parallel.sum_over_ranks(sum(r['points'] for r in results.values()), dist, 'cuda')
    lossless = gop_parallel.sum_over_ranks(sum(0 if (r['lossless'] in (True, None)) else 1 for r in results.values()), dist, 'cuda') == 0
    phase_b = max(wall - phase_a, 1e-9)
    summary = {'frames': args.frames, 'gops': len(groups), 'n_gpus': world, 'schedule': schedule, 'wall_s': round(wall, 4),
               'sec_per_frame': round(wall / args.frames, 6), 'phase_a_s': round(phase_a, 4), 'phase_b_s': round(phase_b, 4),
               'phase_b_efficiency': round(busy_b / (world * phase_b), 4) if len(groups) > 1 else None,
               'ideal_speedup_bound': round(gop_parallel.ideal_speedup(groups, world), 3),
               'bits_per_point': round(bits / max(points, 1.0), 5), 'lossless': bool(lossless) if args.decode else None,
               'inputs_resident_before_t0': bool(stage_all), 'staging_s_this_rank': round(stage_all_s, 2)}
    return summary, results


def resolve_files(args):
    """The frame files of --input-glob / --ori_dir + --ori_dtype (sorted by name, at most --frames of them; args.frames is set to
    their count), or None for the synthetic configs."""
    if getattr(args, 'ori_dir', None) and not args.input_glob:
        args.input_glob = os.path.join(args.ori_dir, '*.' + args.ori_dtype)
    if not args.input_glob:
        return None
    import glob
    files = sorted(glob.glob(args.input_glob))[:args.frames]
    if not files:
        raise ValueError('no file matches %s' % args.input_glob)
    args.frames = len(files)
    return files


def init_dist(local):
    import torch.distributed as dist
    backend = os.environ.get('LINR_BENCH_BACKEND', 'nccl')          # nccl = RCCL over xGMI; gloo only for 1-GPU rehearsals
    # the only collectives are a start-up barrier and the final reductions of wall times: a rank that finishes early waits
    # there for the slowest one, so the watchdog timeout has to cover a whole sequence
    timeout = datetime.timedelta(hours=12)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=timeout)
    else:
        dist.init_process_group(backend, timeout=timeout)
    return dist


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('LINR_BENCH_SINGLE_DEVICE'):          # rehearsal of the multi-rank flow on a 1-GPU box (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dist = init_dist(local) if world > 1 else None
    files = resolve_files(args)
    summary, results = run_sequence_job(args, rank, world, dist, files=files)
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, 'results_rank%d.json' % rank), 'w') as f:
        json.dump({str(k): v for k, v in results.items()}, f, indent=1)
    if rank == 0:
        print(json.dumps(summary))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
