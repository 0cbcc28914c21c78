"""The `rough` leg of bench.py: the non-spherical stress workload (synthetic.rough_figure: a figure-like union of generalised cylinders
with thin parts and a +-8 voxel low-frequency displacement, ~0.75 M points at 10 bit) beside the sphere stand-ins - never the
headline.  What it reports: geometry statistics of the kernel map (taps per row, live taps per 64-row tile, scale sizes), complete
overfits with both training executors (ms/step, ns per row and step against the headline GOP's, bits/point through the codec, lossless
decode) and the reference-trained checkpoint (tests/golden/loot_model_kat.npz, trained by the reference on real loot) through the HIP
forward on this unseen surface.  Counter evidence (bytes per row, L1 hit rate, TA busy): tools/rough_pmc.sh -> profiles/r06_rough_*."""
import os
import time

import numpy as np
import torch

from .common import _popcount32, log


def kmap_stats(gop):
    """K_eff (taps present per row) and live taps per 64-row tile (taps present in ANY row of the tile), row-weighted over the GOP."""
    taps, live, tiles, rows = 0.0, 0.0, 0, 0
    for fr in gop.frames:
        m = fr.nbr_mask[:fr.rows].to(torch.int64) & 0x7FFFFFF
        taps += float(_popcount32(m).sum())
        pad = (-fr.rows) % 64
        mt = torch.cat([m, torch.zeros(pad, dtype=torch.int64, device=m.device)]).view(-1, 64)
        orr = mt[:, 0].clone()
        for j in range(1, 64):
            orr |= mt[:, j]
        live += float(_popcount32(orr).sum())
        tiles += mt.shape[0]
        rows += fr.rows
    return {'k_eff_taps_per_row': round(taps / rows, 2), 'live_taps_per_64_row_tile': round(live / tiles, 2), 'rows': rows, 'tiles': tiles}


def reference_checkpoint_bpp(gop):
    """bits/point (training-forward bits / points) of the reference-trained loot checkpoint on frame 0 through the HIP forward."""
    import ast
    from linr_pcgc_amd import overfit
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'loot_model_kat.npz')
    if not os.path.exists(path) or gop.scale_num != 7:
        return None
    g = np.load(path, allow_pickle=True)
    flat = torch.from_numpy(g['flat'].astype(np.float32))
    sd, off = {}, 0
    for name, shape in zip(g['names'], g['shapes']):
        shape = ast.literal_eval(str(shape))
        cnt = int(np.prod(shape))
        sd[str(name)] = flat[off:off + cnt].view(shape).clone()
        off += cnt
    model = overfit.gen_model(7, 'cpu', seed=1)
    model.load_state_dict(sd)
    model = model.cuda()
    _, bits = model.frame_probs(gop.frames[0])
    return round(float(bits) / gop.point_nums[0], 4)


def rough_leg(epochs, frames, headline_rows, headline_ms, headline_bf16_ms):
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam
    t0 = time.time()
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('loot10_rough', t, 'cuda') for t in range(frames)], None, 64, 'cuda')
    torch.cuda.synchronize()
    out = {'workload': 'synthetic loot10_rough: torso, head, two legs, two thin slanted arms, a thin sheet; +-8 voxel low-frequency '
                       'displacement; '
                       '%d points and %d rows in frame 0, %d scales; 1 GOP of %d frames, %d epochs, seed 8807'
                       % (gop.point_nums[0], gop.frames[0].rows, gop.scale_num, len(gop), epochs),
           'staging_s': round(time.time() - t0, 2), 'kernel_map': kmap_stats(gop),
           'rows_per_scale_frame0': [int(gop.frames[0].row_off[i + 1] - gop.frames[0].row_off[i]) for i in range(gop.frames[0].n_scales)]}
    mean_rows = sum(f.rows for f in gop.frames) / float(len(gop))
    for prec in ('f32', 'bf16'):
        model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
        model.train_precision = prec
        opt = FlatAdam(model)
        init = model.flat_parameters().detach().clone()
        overfit.overfit_gop(model, opt, gop.subset(min(4, len(gop))), 2)                 # clock ramp / first-call costs
        runs = []
        for _ in range(3):
            model.flat_parameters().copy_(init)
            opt.reset()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            info = {}
            e0.record()
            losses = overfit.overfit_gop(model, opt, gop, epochs, info=info)
            e1.record()
            torch.cuda.synchronize()
            runs.append(round(e0.elapsed_time(e1) / (epochs * len(gop)), 4))
        ms = sorted(runs)[1]
        enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8, precision='f32' if prec == 'f32' else 'bf16')
        dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', frames=[0])
        ok = bool(torch.equal(dec[0], torch.as_tensor(gop.infos[0]['ori']).cuda()
                              + torch.tensor(gop.coord_mins[0], device='cuda', dtype=torch.int32)))
        if not ok:
            raise RuntimeError('rough leg: the %s-trained GOP did not decode losslessly' % prec)
        ref_ms = headline_ms if prec == 'f32' else headline_bf16_ms
        out[prec] = {'ms_per_step': ms, 'ms_per_step_runs': runs, 'ns_per_row_step': round(ms * 1e6 / mean_rows, 4),
                     'ns_per_row_step_headline_gop': None if not ref_ms else round(ref_ms * 1e6 / headline_rows, 4),
                     'per_row_cost_over_headline_gop': None if not ref_ms else round((ms / mean_rows) / (ref_ms / headline_rows), 4),
                     'bits_per_point': round(float(enc['bpp']['bpp_all']), 5), 'epoch_loss_bpp': [round(x, 4) for x in losses],
                     'coded_epoch': info.get('coded_epoch'), 'lossless_decode_frame0': ok}
        del enc, dec, model, opt
    out['reference_checkpoint_bits_per_point_frame0'] = reference_checkpoint_bpp(gop)
    out['note'] = ('ms_per_step = median of three complete overfits (same seed); per_row_cost_over_headline_gop compares ns per row and '
                   'step '
                   'with the sphere GOP of the headline on this box; the reference checkpoint was trained by the reference on real loot '
                   '(0.514 bits/point there) and has never seen this surface')
    log('rough leg: %s' % out)
    del gop
    torch.cuda.empty_cache()
    return out
