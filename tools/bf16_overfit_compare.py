"""Complete overfits with the bf16 training executor beside the fp32 one: same GOP, same initialisation seeds, the reference's recipe
(main.py:297-437: Adam, StepLR per frame, lr clamp per epoch, best-epoch checkpoint), then the real streams through the codec -
bits/point as test_utils.py:146-157 counts them - and a lossless decode.  Measurement aid (profiles/r05_bf16_overfit.txt).

  python tools/bf16_overfit_compare.py [--config loot10] [--gop 32] [--epochs 10] [--seeds 8807 1 2] [--codec bf16] [--decode 2]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='loot10')
    ap.add_argument('--gop', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=10)
    ap.add_argument('--seeds', type=int, nargs='+', default=[8807, 1, 2])
    ap.add_argument('--codec', default='bf16', choices=['f32', 'bf16'], help='arithmetic of the coding forward')
    ap.add_argument('--decode', type=int, default=2, help='frames of every GOP to decode and compare with the input')
    args = ap.parse_args()
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam
    t0 = time.time()
    clouds = [synthetic.sequence_frame_device(args.config, t, 'cuda') for t in range(args.gop)]
    gop = overfit.Gop(None, clouds, None, 64, 'cuda')
    torch.cuda.synchronize()
    out = {'config': args.config, 'frames': len(gop), 'rows_frame0': gop.frames[0].rows, 'points_frame0': gop.point_nums[0],
           'epochs': args.epochs, 'codec_precision': args.codec, 'staging_s': round(time.time() - t0, 2), 'runs': []}
    for seed in args.seeds:
        for prec in ('f32', 'bf16'):
            model = overfit.gen_model(gop.scale_num, 'cuda', seed=seed)
            model.train_precision = prec
            opt = FlatAdam(model)
            info = {}
            torch.cuda.synchronize()
            t1 = time.time()
            losses = overfit.overfit_gop(model, opt, gop, args.epochs, info=info)
            torch.cuda.synchronize()
            t2 = time.time()
            enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, precision=args.codec)
            ok = None
            if args.decode:
                todo = list(range(min(args.decode, len(gop))))
                dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, frames=todo)
                ok = all(torch.equal(d, torch.as_tensor(gop.infos[i]['ori']).cuda() + torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32))
                         for d, i in zip(dec, todo))
            out['runs'].append({'seed': seed, 'train': prec, 'loss_per_epoch': [round(x, 4) for x in losses], 'coded_epoch': info['coded_epoch'],
                                'bpp_all': round(enc['bpp']['bpp_all'], 5), 'point_bpp': round(enc['bpp']['point_bpp'], 5),
                                'overfit_s': round(t2 - t1, 3), 'ms_per_step': round((t2 - t1) * 1e3 / (args.epochs * len(gop)), 4), 'lossless': ok})
            print(json.dumps(out['runs'][-1]), flush=True)
    f32 = [r['bpp_all'] for r in out['runs'] if r['train'] == 'f32']
    b16 = [r['bpp_all'] for r in out['runs'] if r['train'] == 'bf16']
    out['mean_bpp_f32'] = round(sum(f32) / len(f32), 5)
    out['mean_bpp_bf16'] = round(sum(b16) / len(b16), 5)
    out['bf16_over_f32'] = round(out['mean_bpp_bf16'] / out['mean_bpp_f32'], 4)
    out['all_lossless'] = all(r['lossless'] in (True, None) for r in out['runs'])
    print(json.dumps({k: v for k, v in out.items() if k != 'runs'}, indent=1))


if __name__ == '__main__':
    main()
