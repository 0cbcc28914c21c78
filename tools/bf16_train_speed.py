"""ms per training step of the bf16 training executor beside the fp32 one, on frames of a synthetic config, with the live
per-class kernel timing of the library (linr_prof_*).  Measurement aid (profiles/r05_bf16_*.txt).

  python tools/bf16_train_speed.py [--config loot10] [--frames 4] [--steps 60] [--classes]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {0: 'f32 fused bwd 8->8', 1: 'f32 conv 8->8 fwd', 2: 'f32 fused dual', 3: 'f32 fused c00', 4: 'f32 head fwd', 5: 'f32 c00|c10 fwd',
         6: 'f32 dual fwd', 7: 'f32 occ conv7', 8: 'f32 head bwd', 9: 'f32 first wgrad', 10: 'f32 lin wgrad', 11: 'f32 sce', 12: 'f32 misc',
         13: 'f32 bwd-data', 17: 'bf16 fused bwd 8->8', 18: 'bf16 fused dual', 19: 'bf16 fused c00', 20: 'bf16 fwd convs', 21: 'bf16 head bwd',
         22: 'bf16 first wgrad', 23: 'bf16 misc'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='loot10')
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--classes', action='store_true', help='per-class kernel table from fully instrumented extra steps')
    args = ap.parse_args()
    from linr_pcgc_amd import _lib, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    L = _lib.lib()
    clouds = [synthetic.sequence_frame_device(args.config, t, 'cuda') for t in range(args.frames)]
    gop = overfit.Gop(None, clouds, None, 64, 'cuda')
    out = {'config': args.config, 'rows': [f.rows for f in gop.frames], 'points': gop.point_nums}
    for prec in ('f32', 'bf16'):
        model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
        model.train_precision = prec
        if prec == 'bf16':
            gop.share_train_bf16_arena()
        opt = FlatAdam(model)
        bits = torch.zeros(1, dtype=torch.float64, device='cuda')

        def run(n):
            for i in range(n):
                j = i % len(gop)
                train_step(model, opt, gop.frames[j], gop.point_nums[j], out=bits)
        run(200)                                        # clock ramp
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(args.steps)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.steps)
        out[prec] = {'ms_per_step': round(best, 4)}
        if args.classes:
            L.linr_prof_mask(0xFFFFFFFF)
            L.linr_prof_enable(1)
            run(32)
            torch.cuda.synchronize()
            L.linr_prof_enable(0)
            tab = {}
            for kind in range(24):
                tot, nl, npass = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
                _lib.check(L.linr_prof_read(kind, ctypes.byref(tot), ctypes.byref(nl), ctypes.byref(npass)), 'linr_prof_read')
                if nl.value:
                    tab[NAMES.get(kind, str(kind))] = {'us_per_step': round(tot.value * 1e3 / 32, 1), 'launches_per_step': nl.value / 32,
                                                       'us_per_pass': round(tot.value * 1e3 / max(npass.value, 1), 2)}
            out[prec]['classes'] = tab
            out[prec]['classes_sum_us'] = round(sum(v['us_per_step'] for v in tab.values()), 1)
            L.linr_prof_mask(3)
        # sanity: the loss after the steps taken
        b = torch.zeros(1, dtype=torch.float64, device='cuda')
        train_step(model, opt, gop.frames[0], gop.point_nums[0], out=b)
        out[prec]['bpp_frame0_now'] = round(float(b) / gop.point_nums[0], 4)
    out['ratio'] = round(out['bf16']['ms_per_step'] / out['f32']['ms_per_step'], 3)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
