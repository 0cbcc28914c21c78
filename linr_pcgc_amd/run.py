"""Sequence driver on the fast path: the overfit -> encode -> decode flow of main.overfit_enc_dec (main.py:69-119) for a
synthetic sequence, one process per GPU (GOP sharding of gop_parallel.py, no data-path collective).

    python -m linr_pcgc_amd.run --config loot10 --frames 64 --gop 32 --first-epoch 10 --others-epoch 10 --out /tmp/linr_out
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m linr_pcgc_amd.run ...
"""
import argparse
import json
import os
import time

import torch

from . import codec, gop_parallel, overfit, ply, synthetic
from .model_core import FlatAdam


def parse():
    ap = argparse.ArgumentParser('linr_pcgc_amd.run')
    ap.add_argument('--config', default='loot10')
    ap.add_argument('--input-glob', default=None, help="PLY / npy frames of a real sequence (sorted by name), e.g. '/data/loot/Ply/*.ply'; replaces --config")
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--gop', type=int, default=32)
    ap.add_argument('--first-epoch', type=int, default=10)
    ap.add_argument('--others-epoch', type=int, default=10)
    ap.add_argument('--learning-rate', type=float, default=0.01)
    ap.add_argument('--gamma', type=float, default=0.992)
    ap.add_argument('--step-size', type=int, default=32)
    ap.add_argument('--min-lr', type=float, default=4e-4)
    ap.add_argument('--decay-rate', type=float, default=1e-4)
    ap.add_argument('--seed', type=int, default=8807)
    ap.add_argument('--out', default='/tmp/linr_out')
    ap.add_argument('--decode', action='store_true', help='decode every GOP again and check it is lossless')
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('LINR_BENCH_SINGLE_DEVICE'):          # rehearsal of the multi-rank flow on a 1-GPU box (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('LINR_BENCH_BACKEND', 'nccl')          # nccl = RCCL over xGMI
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    files = None
    if args.input_glob:
        import glob
        files = sorted(glob.glob(args.input_glob))[:args.frames]
        if not files:
            raise ValueError('no file matches %s' % args.input_glob)
        args.frames = len(files)
    groups = gop_parallel.split_gops(args.frames, args.gop)

    def load_frame(t):
        return ply.read_points(files[t]) if files is not None else synthetic.sequence_frame(args.config, t)

    def make_opt(model):
        return FlatAdam(model, lr=args.learning_rate, weight_decay=args.decay_rate, step_size=args.step_size, gamma=args.gamma)

    def run_gop(group, epochs, ckpt):
        t0 = time.time()
        gop = overfit.Gop(None, [load_frame(t) for t in group], None, 64, 'cuda')
        model = overfit.gen_model(gop.scale_num, 'cuda', seed=args.seed)
        opt = make_opt(model)
        if ckpt is not None:
            overfit.warm_start(model, opt, ckpt)                 # main.py:241-248
        losses = overfit.overfit_gop(model, opt, gop, epochs, args.min_lr)
        enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
        res_dir = os.path.join(args.out, 'result_enc', gop_parallel.gop_name(group))
        codec.write_gop(enc, res_dir)
        ok = None
        if args.decode:
            dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), codec.read_gop(res_dir), 'cuda', workers=4)
            ok = all(torch.equal(d, torch.as_tensor(i['ori']).cuda() + torch.tensor(m, device='cuda', dtype=torch.int32))
                     for d, i, m in zip(dec, gop.infos, gop.coord_mins))
        torch.cuda.synchronize()
        result = {'gop': gop_parallel.gop_name(group), 'frames': len(group), 'epochs': epochs, 'loss': losses,
                  'bpp': enc['bpp'], 'lossless': ok, 'seconds': time.time() - t0, 'rank': rank}
        return model, opt, losses, result

    def first_fn(group):
        model, opt, losses, result = run_gop(group, args.first_epoch, None)
        ck = overfit.checkpoint(model, opt, args.first_epoch - 1, losses[-1])
        ck['result'] = result
        return ck

    def other_fn(group, ckpt):
        return run_gop(group, args.others_epoch, ckpt)[3]

    t0 = time.time()
    results = gop_parallel.run_sequence(groups, os.path.join(args.out, 'output'), first_fn, other_fn, rank, world, dist)
    torch.cuda.synchronize()
    wall = gop_parallel.max_over_ranks(time.time() - t0, dist, 'cuda')
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, 'results_rank%d.json' % rank), 'w') as f:
        json.dump({str(k): v for k, v in results.items()}, f, indent=1)
    if rank == 0:
        print(json.dumps({'frames': args.frames, 'gops': len(groups), 'n_gpus': world, 'wall_s': round(wall, 3),
                          'sec_per_frame': round(wall / args.frames, 4),
                          'ideal_speedup_bound': round(gop_parallel.ideal_speedup(groups, world), 3)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
