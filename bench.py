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
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EPOCHS = 10          # first_epoch / others_epoch of BASELINE config[1]
PROF_EVERY = int(os.environ.get('LINR_BENCH_PROF_EVERY', 8))          # live kernel timing samples every 8th timed step (every step when --steps <= 32)
TABLE_STEPS = 32     # fully instrumented extra steps behind the overfit (per-kernel table)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=320)
    ap.add_argument('--warmup', type=int, default=32)
    ap.add_argument('--ramp-s', dest='ramp_s', type=float, default=1.0,
                    help='seconds of untimed steps before the warm-up steps (clock ramp of a fresh box); 0 disables')
    ap.add_argument('--config', default='loot10', help='synthetic sequence (linr_pcgc_amd.synthetic.CONFIGS)')
    ap.add_argument('--gop', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sequence', action='store_true', help='headline = the whole BASELINE config[2] sequence (strong scaling)')
    ap.add_argument('--no-sequence', action='store_true', help='skip the config[2] sequence leg after the headline')
    ap.add_argument('--seq-frames', type=int, default=300)
    ap.add_argument('--seq-epochs', type=int, default=EPOCHS)
    ap.add_argument('--seq-decode-frames', type=int, default=1, help='frames per GOP decoded and checked in the sequence leg')
    ap.add_argument('--cpu-sample-rows', type=int, default=0, help='0 = whole frame 0')
    return ap.parse_args()


def _time_launches(go, iters):
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


# Kernel classes of the library's live timing (include/linr_hip.h: linr_prof_*) and their ALGORITHMIC bytes per row pass in
# SURVEY.md section 8(d)'s form: conv3(Cin -> Cout) = 4 (Cin + Cout) + 108 (27 int32 neighbour ids), 1x1 / Linear = 4 (Cin + Cout).
# A fused launch counts the row passes it replaces (backward-data + weight gradient of the same convolution = 2 passes).
STEP_ALG_BYTES_PER_ROW = 25476        # forward 8,492 B/row x 3 passes (forward, backward-data, backward-weight)
KERNEL_CLASSES = [
    # kind, name, algorithmic bytes per row and pass (None: not a row-streaming kernel / mixed shapes), kernel-name prefixes in
    # profiles/traffic.json (spaces removed), layer passes a FUSED launch stands for (None: one launch pass = one layer pass)
    (0, 'conv_bwd_wgrad_k<0> fused backward of conv 8->8 (backward-data + weight gradient from one gather)', 2 * 172,
     ['voidconv_bwd_wgrad_k<0,'], 'backward-data + weight gradient of the same convolution: 2 layer passes per launch pass'),
    (1, 'cconv_mfma_k<8,8,fwd> conv 8->8 forward, plain epilogue', 172, ['voidcconv_mfma_k<8,8,false,8,0>'], None),
    (2, 'conv_bwd_wgrad_k<1> fused backward of the two 4->4 convs', 2 * 280, ['voidconv_bwd_wgrad_k<1,'],
     'backward-data + weight gradient of BOTH 4->4 convolutions: 4 layer passes per launch pass'),
    (3, 'conv_bwd_wgrad_k<2> fused backward of conv0_0 8->4 (+ conv1_0 backward-data in the epilogue)', 2 * 156 + 48, ['voidconv_bwd_wgrad_k<2,'],
     'backward-data + weight gradient of conv0_0, backward-data + weight gradient of the 1x1 conv1_0: 4 layer passes per launch pass'),
    (4, 'cconv_mfma_k<8,8,fwd,head> prune conv + head MLP + sigmoid + BCE', 172 + 228, ['voidcconv_mfma_k<8,8,false,8,1>'],
     'conv3 + Linear(8,24) + Linear(24,1) + BCE: 4 layer passes per launch pass'),
    (5, 'cconv_mfma_k<8,4,fwd,pw> conv0_0 + conv1_0', 156 + 48, ['voidcconv_mfma_k<8,4,false,8,2>'], 'conv3 8->4 + 1x1 8->4: 2 layer passes per launch pass'),
    (6, 'cconv_dual44_k<fwd> both 4->4 convs + conv1_2 + residual', 280 + 32, ['voidcconv_dual44_k<false>'],
     'two conv3 4->4 + 1x1 4->4: 3 layer passes per launch pass'),
    (7, 'occ_conv7_k first convs of the 7 outter blocks (one gather; 7 layer passes)', 156, ['occ_conv7_k'],
     'seven first convolutions from one gather: the 7 layer passes are counted as passes of this launch'),
    (8, 'head_bwd_k head MLP backward (data + weights)', 2 * 228, ['head_bwd_k'], 'backward-data + weight gradients of both Linear layers: 4 layer passes per launch pass'),
    (9, 'occ_wgrad7_k weight gradients of the first convs of the 7 outter blocks (one gather; 7 layer passes)', 156, ['occ_wgrad7_k', 'voidspconv_wgrad_t_k'],
     'seven weight gradients from one gather: the 7 layer passes are counted as passes of this launch'),
    (10, 'xtg_wgrad_k pointwise weight gradients', None, ['voidxtg_wgrad_k'], None),
    (11, 'sce_fwd_k / sce_bwd_k scale context', None, ['sce_fwd_k', 'sce_bwd_all_k'], None),
    (12, 'sum8_k, wgrad_reduce_k, sce_emb_grad, adam_k, bits finish', None, ['sum8_k', 'wgrad_reduce_k', 'sce_emb_grad_all_k', 'adam_k', 'bce_bits_finish_k'], None),
    (13, 'stand-alone backward-data convolutions (schedules without the fused backward)', 172,
     ['voidcconv_mfma_k<8,8,true', 'voidcconv_mfma_k<4,8,true', 'voidcconv_dual44_k<true>'], None),
]


def _popcount32(t):
    """Set bits per element of an int32 tensor (the 27-bit masks of the compressed kernel map)."""
    v = t.to(torch.int64) & 0xFFFFFFFF
    v = v - ((v >> 1) & 0x55555555)
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333)
    v = (v + (v >> 4)) & 0x0F0F0F0F
    return (v * 0x01010101 >> 24) & 0xFF


def _read_prof(L, _lib):
    import ctypes
    out = {}
    for kind, *_ in KERNEL_CLASSES:
        tot, nl, npass = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.linr_prof_read(kind, ctypes.byref(tot), ctypes.byref(nl), ctypes.byref(npass)), 'linr_prof_read')
        out[kind] = (tot.value, nl.value, npass.value)
    return out


def load_traffic(name='traffic.json'):
    """profiles/traffic.json (fp32 executor) / traffic_bf16.json (bf16 training executor): HBM bytes of every kernel of a training step
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_probe.py (tools/traffic_pmc.sh; FETCH_SIZE doubled
    for gfx950).  STORED numbers of the build named inside the file (`library`), not a measurement of this run: traffic_source() says so."""
    tpath = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(tpath):
        return {}
    try:
        return json.load(open(tpath))
    except Exception:
        return {}


def traffic_source(traffic, name='traffic.json'):
    """One sentence for the bench line: where the stored counter bytes came from and which library build they describe."""
    lib = traffic.get('library') or {}
    here = os.path.join(ROOT, 'linr_pcgc_amd', 'liblinr_hip.so')
    now = time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime(os.path.getmtime(here))) if os.path.exists(here) else None
    return ('stored counters of profiles/%s (collected %s on the library built %s, %s bytes; the library running now: built %s, %s bytes)'
            % (name, traffic.get('collected_utc'), lib.get('built_utc'), lib.get('bytes'), now, os.path.getsize(here) if now else None))


def counter_bytes_per_step(traffic, prefixes, mean_rows):
    """Sum of the counter bytes per training step over the kernel names with one of the prefixes, scaled from the probe's frame to
    this GOP's mean row count.  None when the counter file does not hold any of them."""
    kernels = traffic.get('kernels', {})
    scale = (mean_rows / float(traffic['rows'])) if traffic.get('rows') else 1.0
    tot, hit = 0.0, False
    for k, v in kernels.items():
        if any(k.replace(' ', '').startswith(p) for p in prefixes) and 'bytes_per_step' in v:
            tot += v['bytes_per_step'] * scale
            hit = True
    return tot if hit else None


def kernel_table(table_prof, table_steps, mean_rows, ms_per_step, traffic):
    """Per-class view of one training step from the fully instrumented pass (every launch bracketed by an event pair; run outside
    the timed region because ~30 event pairs per step cost ~3 % of it).  Two byte figures per class: the ALGORITHMIC bytes of
    SURVEY.md section 8(d) (a fused launch is credited with every layer pass it stands for - `fused` says which - so its
    `frac_alg_bookkeeping` can exceed 1) and the COUNTER bytes the launches really moved (`frac_counter` = counter bytes / time /
    8 TB/s: always <= 1, and what says how far the memory system is from its limit)."""
    rows, covered, counter_total, counter_missing = [], 0.0, 0.0, []
    for kind, name, alg, prefixes, fused in KERNEL_CLASSES:
        tot_ms, launches, passes = table_prof[kind]
        if launches == 0:
            continue
        us_step = tot_ms * 1e3 / table_steps
        covered += us_step
        e = {'kernel': name, 'launches_per_step': round(launches / table_steps, 2), 'row_passes_per_step': round(passes / table_steps, 2),
             'us_per_step': round(us_step, 1)}
        if alg is not None and passes > 0:
            gbs = (passes / table_steps) * mean_rows * alg / (us_step * 1e-6) / 1e9
            e.update({'alg_bytes_per_row_pass': alg, 'alg_gbs': round(gbs, 1), 'frac_alg_bookkeeping': round(gbs / HBM_PEAK_GBS, 4)})
        if fused:
            e['fused'] = fused
        cb = counter_bytes_per_step(traffic, prefixes, mean_rows)
        if cb is not None:
            counter_total += cb
            e.update({'counter_bytes_per_step': int(cb), 'counter_gbs': round(cb / (us_step * 1e-6) / 1e9, 1),
                      'frac_counter': round(cb / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
        else:
            counter_missing.append(name.split(' ')[0])
        rows.append(e)
    return rows, covered, counter_total, counter_missing


def kernel_roofline(gop, live, table_prof, table_steps, ms_per_step):
    """Dominant kernel = the top line of the rocprofv3 kernel statistics of this command (profiles/): conv_bwd_wgrad_k<0>, the
    fused backward of the 8->8 convolutions (prune convs, tail convs, block_in's first conv: 17 convolution backward passes per
    step in 3 launches, each pass = backward-data AND weight gradient from one gather).  `avg_launch_us` is measured LIVE over
    the timed region: the library brackets every launch of the kernel inside the training steps with a HIP event pair on the
    launch stream (linr_prof_enable / linr_prof_read), so it is the number rocprofv3's AverageNs of the same command must agree
    with.  Algorithmic bytes per launch: groups x rows x 2 x (4 (8 + 8) + 108) - the two row passes the
    launch replaces.  `traffic`: HBM bytes per launch of the executor's 8-group launch from separate --pmc passes
    (profiles/traffic.json, tools/traffic_pmc.sh).  `step`: the whole step against SURVEY's 25,476 B/row; `kernels`: every
    kernel class of a step from the fully instrumented pass."""
    mean_rows = sum(fr.rows for fr in gop.frames) / len(gop.frames)
    traffic_all = load_traffic()
    traffic = traffic_all.get('kernels', {})
    # taps present per row (K_eff): the popcount of the compressed map's 27-bit masks, row-weighted over the GOP
    k_eff = float(sum(float(_popcount32(fr.nbr_mask[:fr.rows]).sum()) for fr in gop.frames) / sum(fr.rows for fr in gop.frames))

    def entry(kind, name, alg_per_pass, flops_per_pass, traffic_key):
        tot_ms, launches, passes = live[kind]
        if launches == 0:
            return None
        dur_s = tot_ms / 1e3 / launches
        ppl = passes / launches
        alg = ppl * mean_rows * alg_per_pass
        achieved = alg / dur_s / 1e9
        tr = None
        for k, v in traffic.items():
            if k.replace(' ', '').startswith(traffic_key):
                tr = v['bytes_per_dispatch']
        tflops = ppl * mean_rows * flops_per_pass / dur_s / 1e12       # dense-27 flops the kernel executes on the matrix cores
        return {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': tr,
                'frac_counter': None if tr is None else round(tr / dur_s / 1e9 / HBM_PEAK_GBS, 4), 'kernel': name,
                'mfma_f32_view': {'achieved_tflops': round(tflops, 1), 'peak_tflops': 157.3, 'frac': round(tflops / 157.3, 4),
                                  # SURVEY 8(d) counts 2 K_row Cin Cout: only the taps that exist.  The kernels issue all 27.
                                  'k_eff_taps_per_row': round(k_eff, 2), 'useful_tflops': round(tflops * k_eff / 27.0, 1),
                                  'useful_frac': round(tflops * k_eff / 27.0 / 157.3, 4),
                                  # v_mfma_f32_4x4x1 issues every 9.5-10 cycles, not 8 (profiles/r03_issue_probe.txt): what a
                                  # stream of nothing but these instructions reaches
                                  'issue_ceiling_tflops_4x4x1': 119.0, 'frac_of_issue_ceiling': round(tflops / 119.0, 4)},
                'launches_timed': int(launches), 'passes_per_launch': round(ppl, 3), 'rows_per_pass': round(mean_rows, 1),
                'alg_bytes_per_launch': int(alg), 'avg_launch_us': round(dur_s * 1e6, 2)}

    roof = entry(0, KERNEL_CLASSES[0][1], 2 * 172, 2 * 2 * 27 * 8 * 8, 'voidconv_bwd_wgrad_k<0,3>')
    if roof is None:          # debug switches: the executor did not run the fused kernel
        roof = {'bound': 'hbm', 'achieved': None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None, 'traffic': None,
                'kernel': 'conv_bwd_wgrad_k<0> not launched (LINR_FUSED_BWD=0?)'}
    if roof.get('traffic') is not None:
        roof['traffic_note'] = ('HBM bytes of the 8-group tail-convolution launch (conv_bwd_wgrad_k<0,3>); live launches average '
                                '%.2f groups; %s' % (roof.get('passes_per_launch', 0.0), traffic_source(traffic_all)))
    roof['conv'] = entry(1, KERNEL_CLASSES[1][1], 172, 2 * 27 * 8 * 8, 'voidcconv_mfma_k<8,8,false,8,0>')
    step_gbs = STEP_ALG_BYTES_PER_ROW * mean_rows / (ms_per_step * 1e-3) / 1e9
    roof['step'] = {'alg_bytes_per_row': STEP_ALG_BYTES_PER_ROW, 'rows': round(mean_rows, 1), 'ms_per_step': round(ms_per_step, 4),
                    'achieved': round(step_gbs, 1), 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'frac': round(step_gbs / HBM_PEAK_GBS, 4)}
    if table_prof is not None:
        rows, covered, counter_total, counter_missing = kernel_table(table_prof, table_steps, mean_rows, ms_per_step, traffic_all)
        roof['kernels'] = rows
        if counter_total > 0:
            roof['step'].update({'traffic': int(counter_total), 'traffic_over_algorithmic': round(counter_total / (STEP_ALG_BYTES_PER_ROW * mean_rows), 4),
                                 'frac_counter': round(counter_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 'traffic_note': 'sum of the counter bytes of every kernel class of a step; ' + traffic_source(traffic_all)
                                                 + ('; no counters for: ' + ', '.join(counter_missing) if counter_missing else '')})
        roof['kernels_note'] = ('%d extra steps with EVERY launch bracketed by a HIP event pair, outside the timed region (state '
                                'saved and restored); sum %.1f us = %.3f of the un-instrumented step'
                                % (table_steps, covered, covered / (ms_per_step * 1e3)))
    return roof


def cpu_baseline(model_sd, gop_info, point_num, sample_rows):
    """The CPU oracle ("port": ME/torchac are not installable, the reference has no CPU path) on the host cores:
    one overfit step (forward + autograd backward + Adam) + one inference forward on frame 0.
    Also returns the oracle's bits and per-tensor gradients of that step for the full-size parity check."""
    from oracle import network as onet
    scales = []
    for s in gop_info['all_input_info']:
        scales.append({'coord': s['coord'].cpu().numpy(), 'occ': s['occ'].cpu().numpy(),
                       'offset_tensor': s['offset_tensor'].cpu().numpy(), 'scale_idx': s['scale_idx']})
    rows = sum(len(s['coord']) for s in scales)
    t0 = time.time()
    tsc = onet.to_torch_scales(scales)          # builds the kernel maps (oracle.octree.neighbour_table: sorted-key searches in numpy)
    t_kmap = time.time() - t0
    sd = {k: v.clone().requires_grad_() for k, v in model_sd.items()}
    flat_p = torch.cat([v.detach().reshape(-1) for v in sd.values()])
    m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
    t0 = time.time()
    bits = onet.frame_bits(sd, tsc)
    (bits / point_num).backward()
    g = torch.cat([t.grad.reshape(-1) for t in sd.values()])
    grads = {k: t.grad.detach().clone() for k, t in sd.items()}
    onet.adam_step(flat_p, g.clone(), m, v, 1, 0.01)
    t_step = time.time() - t0
    from oracle import ac as oac
    with torch.no_grad():
        t0 = time.time()
        sdd = {k: v.detach() for k, v in sd.items()}
        outs = [onet.forward_scale(sdd, s) for s in tsc]
        t_fwd = time.time() - t0
        # the arithmetic-coder feed of encode (models/upsample.py:224-237): 8 streams per scale through the oracle's plain-C
        # restatement of torchac's coder, one thread (torchac's own encoder is serial too)
        t0 = time.time()
        ac_bytes = 0
        for s, o in zip(scales, outs):
            for k in range(8):
                ac_bytes += len(oac.encode_binary(o['probs'][k].reshape(-1).numpy(), s['occ'][:, k].astype(np.uint8)))
        t_ac = time.time() - t0
    out = {'value': round(EPOCHS * t_step + t_fwd + t_ac, 3), 'unit': 's/frame', 'cores': torch.get_num_threads(),
           'kind': 'port',
           'sample': '1 overfit step (%.2f s) + 1 forward (%.2f s) + range coding of its %d symbols (%.3f s, 1 thread, %d bytes) of '
                     'frame 0 (%d rows), x%d epochs of the step' % (t_step, t_fwd, 8 * rows, t_ac, ac_bytes, rows, EPOCHS),
           'train_step_s': round(t_step, 3), 'forward_s': round(t_fwd, 3), 'ac_s': round(t_ac, 4),
           'kernel_map_s': round(t_kmap, 3),          # once per frame, outside `value` like the GPU side's staging
           'bits_frame0_init': float(bits.detach())}
    return out, float(bits.detach()), grads


# full-size parity (frame 0, 336 k rows, initial parameters): HIP forward/backward against the oracle step the CPU baseline
# runs anyway.  bits: relative 1e-5; gradients PER TENSOR: max |d| <= 1e-3 * max |g| of that tensor
# + 1e-9 (two fp32 evaluations with 336 k-row sums in different orders and heavy cancellation; measured worst 1.2e-4 -
# the float64-anchored criterion lives in tests/test_gpu_parity.py, where the oracle is cheap enough to run twice).
PARITY_BITS_RTOL = 1e-5
PARITY_GRAD_RTOL = 3e-4


def full_size_parity(model_sd, frame, point_num, oracle_bits, oracle_grads, scale_num):
    from linr_pcgc_amd import engine, overfit
    model = overfit.gen_model(scale_num, 'cuda')
    model.load_state_dict(model_sd)
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    engine.net_forward(frame, model.flat_parameters(), 0, 8, None, bits)
    flat_g = torch.zeros_like(model.flat_parameters())
    engine.net_backward(frame, model.flat_parameters(), flat_g, 1.0 / float(point_num))
    torch.cuda.synchronize()
    got_bits = float(bits)
    worst, worst_name, off = 0.0, '', 0
    flat_g = flat_g.cpu()
    for name, p in model.state_dict().items():
        n = p.numel()
        g_hip = flat_g[off:off + n].view(p.shape)
        g_ref = oracle_grads[name]
        off += n
        gmax = float(g_ref.abs().max())
        err = float((g_hip - g_ref).abs().max())
        rel = err / (gmax + 1e-30) if gmax > 0 else (0.0 if err <= 1e-9 else float('inf'))
        if err > 1e-9 and rel > worst:
            worst, worst_name = rel, name
    bits_rel = abs(got_bits - oracle_bits) / abs(oracle_bits)
    ok = bits_rel <= PARITY_BITS_RTOL and worst <= PARITY_GRAD_RTOL
    return {'ok': bool(ok), 'bits_hip': got_bits, 'bits_oracle': oracle_bits, 'bits_rel_err': bits_rel,
            'grad_worst_rel_err_per_tensor': worst, 'grad_worst_tensor': worst_name, 'tensors': len(oracle_grads),
            'tolerance': {'bits_rel': PARITY_BITS_RTOL, 'grad_rel_to_own_tensor_max': PARITY_GRAD_RTOL}}


def steps_done_so_far(steps, rest, total):
    """True when the run covered exactly one complete overfit (the best-epoch bookkeeping is per overfit)."""
    return steps + rest == total


def log(msg):
    if int(os.environ.get('RANK', 0)) == 0:
        print('[bench %7.1fs] %s' % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def host_threads():
    """CPU threads this process may really use (the GPU box gives one GPU a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def sequence_leg(args, rank, world, dist, stage_all=True):
    """BASELINE config[2] for real: seq_frames frames in GOPs of args.gop, GOP 0 from scratch on rank 0, the other GOPs
    warm-started from its checkpoint and dealt over the ranks (static longest-first deal so that every input is staged in
    HBM before the timed region starts), each GOP overfitted, encoded to files and spot-decoded.  Strong scaling: the
    work is fixed, `sec_per_frame` = whole-sequence wall / frames."""
    import shutil
    import tempfile
    from linr_pcgc_amd import run as seq_run
    out_dir = None
    if rank == 0:
        out_dir = tempfile.mkdtemp(prefix='linr_seq_')
    if dist is not None:
        box = [out_dir]
        dist.broadcast_object_list(box, src=0)
        out_dir = box[0]
    sargs = seq_run.parse(['--config', args.config, '--frames', str(args.seq_frames), '--gop', str(args.gop),
                           '--first-epoch', str(args.seq_epochs), '--others-epoch', str(args.seq_epochs), '--out', out_dir,
                           '--decode'])
    try:
        summary, _ = seq_run.run_sequence_job(sargs, rank, world, dist, stage_all=stage_all, decode_frames=args.seq_decode_frames)
    finally:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(out_dir, ignore_errors=True)
    summary['workload'] = ('BASELINE config[2] stand-in: synthetic %s, %d frames, GOP %d, first_epoch=others_epoch=%d, GOP 0 serial '
                           'prefix then GOPs over %d GPU(s), no collective; %d frame(s) per GOP decoded and compared'
                           % (args.config, args.seq_frames, args.gop, args.seq_epochs, world, args.seq_decode_frames))
    return summary


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
    while time.time() - t_ramp < 1.5:                       # clock ramp on the kernels that are about to be timed (the legs before this one are host-bound)
        for _ in range(64):
            train_step(model, opt, gop.frames[i % len(gop)], gop.point_nums[i % len(gop)], out=bits)
            i += 1
        torch.cuda.synchronize()
    steps = epochs * len(gop)
    ms, runs = None, []
    for _ in range(2):                                      # two complete overfits from the same seed (bit-identical trajectories): the faster one is reported
        model.flat_parameters().copy_(init)
        opt.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        info = {}
        e0.record()
        losses = overfit.overfit_gop(model, opt, gop, epochs, info=info)      # the complete overfit, un-instrumented: ms_per_step, bits/point
        e1.record()
        torch.cuda.synchronize()
        runs.append(round(e0.elapsed_time(e1) / steps, 4))
        ms = runs[-1] if ms is None else min(ms, runs[-1])
    # launch durations of the dominant kernel class, live (event pairs on the launch stream), from 64 more steps of a scratch copy of the
    # trained state - outside the timed overfit, whose model is what gets coded below
    snap = (model.flat_parameters().detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t, opt.t_scale.copy(), opt.lr, opt.sched_steps)
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
    ok = all(bool(torch.equal(dec[i], torch.as_tensor(gop.infos[i]['ori']).cuda() + torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32)))
             for i in range(nd))
    out.update({'ms_per_step': round(ms, 4), 'steps': steps, 'ms_per_step_runs': runs,
                'note': 'the complete %d-epoch overfit incl. its per-epoch host reads of the loss (HIP events); the faster of two runs from the same seed' % epochs,
                'epoch_loss_bpp': [round(x, 4) for x in losses], 'coded_epoch': info.get('coded_epoch'),
                'bits_per_point': round(float(enc['bpp']['bpp_all']), 5), 'codec': 'bf16 features / uint8 weight codes', 'lossless_decode_frames0to1': ok})
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
            tr_note = 'HBM bytes per 8-group launch of bbwd_k<0,3> / 8 x the mean groups per launch; ' + traffic_source(tb, 'traffic_bf16.json')
        out['roofline'] = {'kernel': 'bbwd_k<0>: fused backward-data + weight gradient of the convolutions 8->8 (17 of a step\'s 33 backward row passes, 3 launches)',
                           'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(achieved / 8000.0, 4),
                           'alg_bytes_per_row_pass': alg_row_pass, 'row_passes_per_fused_group': 2, 'mean_groups_per_launch': round(groups, 3),
                           'mean_launch_us': round(us_launch, 2), 'us_per_group_pass': round(tot.value * 1e3 / max(npass.value, 1), 2),
                           'launches_sampled': int(nl.value), 'traffic': tr_launch, 'traffic_note': tr_note,
                           'frac_note': 'algorithmic bytes of SURVEY 8(d): each of the two row passes a fused launch replaces is priced with a 108-byte '
                                        'neighbour table, which the kernel streams as 40 bytes and once - frac can exceed 1; frac_counter prices the bytes '
                                        'the memory system moved',
                           'frac_counter': None if tr_launch is None else round(tr_launch / (us_launch * 1e-6) / 1e9 / 8000.0, 4),
                           'step': {'alg_bytes_per_row': 20514, 'achieved': round(20514 * mean_rows / (ms * 1e-3) / 1e9, 1),
                                    'frac': round(20514 * mean_rows / (ms * 1e-3) / 1e9 / 8000.0, 4),
                                    'note': 'SURVEY 8(d) at 2-byte features: 3 passes x (1,654 B features + 5,184 B neighbour table) per row'}}
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
            res['ms_per_step_' + prec] = round(_time_launches(lambda: train_step(m4, o4, g4.frames[0], g4.point_nums[0], out=bits), 30) * 1e3, 4)
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
            m64 = overfit.gen_model(g64.scale_num, 'cuda', seed=8807)
            m64.train_precision = 'bf16'
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
            d64 = codec.decode_gop(overfit.gen_model(g64.scale_num, 'cuda'), e64, 'cuda', frames=[0])
            ok64 = bool(torch.equal(d64[0], torch.as_tensor(g64.infos[0]['ori']).cuda() + torch.tensor(g64.coord_mins[0], device='cuda', dtype=torch.int32)))
            out['config4_gop64'] = {'workload': 'BASELINE config[4] stand-in: synthetic owlii11 (11-bit sphere shell, %d points and %d rows in frame 0, %d scales), ONE GOP of 64 '
                                                'frames, %d epochs of bf16 training, bf16 / uint8-weight codec' % (g64.point_nums[0], g64.frames[0].rows, g64.scale_num, epochs),
                                    'encode_sec_per_frame': round((t2 - t0) / 64.0, 5), 'overfit_s': round(t1 - t0, 3), 'codec_s': round(t2 - t1, 3),
                                    'ms_per_step': round((t1 - t0) * 1e3 / (epochs * 64), 4), 'staging_s': round(stage_s, 2),
                                    'bits_per_point': round(float(e64['bpp']['bpp_all']), 5), 'epoch_loss_bpp': [round(x, 4) for x in l64],
                                    'coded_epoch': info64.get('coded_epoch'), 'lossless_decode_frame0': ok64,
                                    'note': 'encode = overfit + codec of the whole GOP on this one GPU, inputs resident; the first codec call of this GOP size (it sizes the pinned staging ring)'}
            del g64, m64, o64, e64, d64
        except Exception as e:
            out['config4_gop64'] = {'error': repr(e)}
    torch.cuda.empty_cache()
    return out



class Headline:
    """The timed region: K steps of the per-GOP overfit (main.py:297-321) on this rank's GOP, then the rest of the complete overfit.
    Holds everything the timed loop touches (created before the ramp): the loss accumulators, the per-step events, the device-side
    best-epoch snapshot (the reference codes with the epoch of the lowest mean loss, main.py:413-426,440-451; tracked on the device
    so that the loop never waits for the host)."""

    def __init__(self, args, rank, L, _lib):
        from linr_pcgc_amd import overfit, synthetic
        from linr_pcgc_amd.model_core import FlatAdam, train_step
        self.args, self.L, self._lib, self.train_step = args, L, _lib, train_step
        # rank r owns GOP r of the sequence: frames [gop*r, gop*(r+1))  (GOPs are independent: no collective)
        t_setup = time.time()
        clouds = [synthetic.sequence_frame_device(args.config, rank * args.gop + t, 'cuda') for t in range(args.gop)]
        self.gop = gop = overfit.Gop(None, clouds, None, 64, 'cuda')
        del clouds
        self.model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
        self.init_sd = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        self.init_flat = self.model.flat_parameters().detach().clone()           # device copy: the reset before t0 is one D2D copy
        self.setup_s = time.time() - t_setup
        log('setup done: %d frames, frame0 %d points / %d rows, %d scales' % (len(gop), gop.point_nums[0], gop.frames[0].rows, gop.scale_num))
        self.opt = FlatAdam(self.model)
        self.total_steps = EPOCHS * len(gop)
        self.prof_every = 1 if args.steps <= 32 else PROF_EVERY
        L.linr_prof_mask(3)                                       # timed region: the dominant kernel and the forward conv only
        _lib.check(L.linr_prof_enable(1), 'linr_prof_enable')     # creates the event pairs ...
        L.linr_prof_enable(0)                                     # ... and stops; sampled steps switch it on (mode 2)
        self.acc = torch.zeros(len(gop), dtype=torch.float64, device='cuda')
        self.pns = torch.tensor([float(pn) for pn in gop.point_nums], dtype=torch.float64, device='cuda')
        epoch_end = (self.acc / self.pns).sum()                   # loads the torch kernels the epoch end uses
        del epoch_end
        self.step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        self.epoch_loss = []
        self.flat = self.model.flat_parameters()
        self.best = {'loss': torch.full((), float('inf'), dtype=torch.float64, device='cuda'),
                     'epoch': torch.full((), -1, dtype=torch.int64, device='cuda'), 'p': self.flat.detach().clone(),
                     'm': self.opt.exp_avg.clone(), 'v': self.opt.exp_avg_sq.clone(), 'meta': []}

    def best_reset(self):
        self.best['loss'].fill_(float('inf'))
        self.best['epoch'].fill_(-1)
        self.best['meta'].clear()

    def best_offer(self, l):
        best, opt = self.best, self.opt
        better = l < best['loss']
        torch.where(better, self.flat.detach(), best['p'], out=best['p'])
        torch.where(better, opt.exp_avg, best['m'], out=best['m'])
        torch.where(better, opt.exp_avg_sq, best['v'], out=best['v'])
        best['epoch'].copy_(torch.where(better, torch.full_like(best['epoch'], len(best['meta'])), best['epoch']))
        best['loss'].copy_(torch.minimum(best['loss'], l))
        best['meta'].append((opt.t, opt.t_scale.copy(), opt.lr, opt.sched_steps))

    def body(self, i, sample):
        """One iteration of the timed loop - warm-up and ramp run exactly this."""
        gop = self.gop
        j = i % len(gop)
        if sample:
            self.L.linr_prof_enable(2)
        self.train_step(self.model, self.opt, gop.frames[j], gop.point_nums[j], out=self.acc[j:j + 1])      # bits of frame j into its own slot
        if sample:
            self.L.linr_prof_enable(0)
        if j == len(gop) - 1:
            l = (self.acc / self.pns).sum()                 # like overfit.overfit_gop: per-epoch loss, no per-step torch kernels
            self.best_offer(l)                              # before the clamp, as the reference saves (main.py:413-437)
            self.opt.clamp_lr(4e-4)
            self.epoch_loss.append(l)
            self.acc.zero_()

    def run(self, barrier, dist):
        """Ramp + W warm-up steps, reset in place, the K timed steps, then the rest of the complete overfit (second timed region)."""
        args, L = self.args, self.L
        # a fresh box starts at idle clocks (sclk level 1): ramp the device with ~1 s of the same steps before the W warm-up
        # steps, otherwise the first few hundred timed steps run ~10 % slow (measured: 3.22 vs 2.92 ms/step)
        t_ramp, i_ramp = time.time(), 0
        while time.time() - t_ramp < args.ramp_s:
            for _ in range(32):
                self.body(i_ramp, i_ramp % self.prof_every == 0)
                i_ramp += 1
            torch.cuda.synchronize()
        for i in range(args.warmup):
            self.body(i, i % self.prof_every == 0)
        # reset to the seeded initialisation IN PLACE (one D2D copy + three memsets on the stream; nothing is allocated and
        # the host does not wait), drop the warm-up's samples
        self.model.flat_parameters().copy_(self.init_flat)
        self.opt.reset()
        self.acc.zero_()
        self.epoch_loss.clear()
        self.best_reset()
        barrier()
        L.linr_prof_enable(1)                                     # clears the records (the events are reused, none is created)
        L.linr_prof_enable(0)
        log('warm-up done (%d ramp + %d warm-up steps)' % (i_ramp, args.warmup))
        barrier()
        t0 = time.time()
        self.step_ev[0].record()
        for i in range(args.steps):
            self.body(i, i % self.prof_every == 0)
            self.step_ev[i + 1].record()
        barrier()
        elapsed = time.time() - t0
        per_step_ms = [self.step_ev[i].elapsed_time(self.step_ev[i + 1]) for i in range(args.steps)]
        self.live = _read_prof(L, self._lib)
        # carry the overfit on to its full length (second timed region) so that bits/point and value describe one training
        rest = max(0, self.total_steps - args.steps)
        barrier()
        t1 = time.time()
        for i in range(args.steps, args.steps + rest):
            self.body(i, False)
        # leave model and optimiser in the state of the best epoch (what the reference's model.pth holds) - inside the timed region
        best, opt = self.best, self.opt
        self.coded_epoch = int(best['epoch'])
        if 0 <= self.coded_epoch < len(best['meta']) and steps_done_so_far(args.steps, rest, self.total_steps):
            self.flat.detach().copy_(best['p'])
            opt.exp_avg.copy_(best['m'])
            opt.exp_avg_sq.copy_(best['v'])
            opt.t, opt.t_scale, opt.lr, opt.sched_steps = (best['meta'][self.coded_epoch][0], best['meta'][self.coded_epoch][1].copy(),
                                                           best['meta'][self.coded_epoch][2], best['meta'][self.coded_epoch][3])
        barrier()
        rest_s = time.time() - t1
        if dist is not None:
            t = torch.tensor([elapsed, rest_s], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, rest_s = float(t[0]), float(t[1])
        self.elapsed, self.rest_s = elapsed, rest_s
        self.ms_per_step = elapsed * 1e3 / args.steps
        self.steps_done = args.steps + rest
        self.full_overfit_s = (elapsed + rest_s) * (self.total_steps / float(self.steps_done))       # steps > total: scaled back to one overfit
        self.losses = [float(x) / len(self.gop) for x in self.epoch_loss]
        srt = sorted(per_step_ms)
        self.step_stats = {'min': round(srt[0], 4), 'median': round(srt[len(srt) // 2], 4), 'max': round(srt[-1], 4),
                           'first8': [round(x, 3) for x in per_step_ms[:8]], 'sum_over_wall': round(sum(per_step_ms) / (elapsed * 1e3), 4)}
        log('timed %d steps: %.3f ms/step (events: min %.3f median %.3f max %.3f); full overfit %d steps %.3f s; epoch losses %s'
            % (args.steps, self.ms_per_step, srt[0], srt[len(srt) // 2], srt[-1], self.steps_done, elapsed + rest_s,
               ['%.4f' % x for x in self.losses]))

    def kernel_table_leg(self):
        """Per-kernel table: TABLE_STEPS more steps with every launch of a step bracketed by an event pair (outside every timed region;
        parameters and optimiser state are saved and put back, so the codec leg codes the model of the complete overfit)."""
        L, opt = self.L, self.opt
        snap = (self.model.flat_parameters().detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t, opt.t_scale.copy(),
                opt.lr, opt.sched_steps)
        n_loss = len(self.epoch_loss)
        L.linr_prof_mask(0xFFFFFFFF)
        L.linr_prof_enable(1)
        for i in range(TABLE_STEPS):
            self.body(i, False)
        L.linr_prof_enable(0)
        torch.cuda.synchronize()
        table_prof = _read_prof(L, self._lib)
        L.linr_prof_mask(3)
        self.model.flat_parameters().copy_(snap[0])
        opt.exp_avg.copy_(snap[1])
        opt.exp_avg_sq.copy_(snap[2])
        opt.t, opt.t_scale, opt.lr, opt.sched_steps = snap[3], snap[4], snap[5], snap[6]
        self.acc.zero_()
        del self.epoch_loss[n_loss:]
        torch.cuda.synchronize()
        return table_prof


def codec_leg(h, rank, dist, barrier):
    """Outside the K timed steps: model compression + per-frame forward + D2H + AC + the bitstream files of encoder.py:13-18,81-118
    (T_write of the metric).  Timed twice: the FIRST call of a process pays for the pinned staging ring (hipHostMalloc of ~54 MB), the
    coder's thread pool and first-use kernels - one-time costs that a single 32-frame GOP would otherwise be charged with (3.0-3.7 vs
    1.5 ms/frame); like the W warm-up steps of the overfit it is reported (`codec_first_call`) but `value` uses the second,
    steady-state call - what every later GOP of a sequence costs."""
    import shutil
    import tempfile
    from linr_pcgc_amd import codec, overfit
    gop, model = h.gop, h.model
    model_ori = overfit.gen_model(gop.scale_num, 'cuda')
    out_dir = tempfile.mkdtemp(prefix='linr_bench_rank%d_' % rank)
    barrier()
    t0 = time.time()
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    codec.write_gop(enc, out_dir)
    barrier()
    codec_cold_s = time.time() - t0
    shutil.rmtree(out_dir, ignore_errors=True)
    barrier()
    t0 = time.time()
    enc = codec.encode_gop(model, model_ori, gop, 8)
    codec.write_gop(enc, out_dir)
    barrier()
    codec_s = time.time() - t0
    shutil.rmtree(out_dir, ignore_errors=True)
    if dist is not None:
        t = torch.tensor([codec_s], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        codec_s = float(t)
    log('encode leg: %.3f s/frame, bpp %.4f' % (codec_s / len(gop), enc['bpp']['bpp_all']))
    return enc, codec_s, codec_cold_s


def decode_leg(h, enc):
    """Decode check (outside the metric): frames 0..3 from the streams alone, 4 frames in flight (the first call also pays for the pinned
    staging buffers, so the timing is taken on a second pass); then the decoder's two other operating points: one frame alone
    (latency: 56 dependent stage forwards + range decoding of ~2.7 M symbols on one host thread) and 8 frames in flight
    (throughput); the once-per-GOP part of decode_gop (model.bin -> parameters, coarsest coordinates) is reported on its own."""
    from linr_pcgc_amd import codec, overfit
    gop = h.gop
    nd = min(4, len(gop))
    dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', frames=list(range(nd)), workers=nd)
    lossless = True
    for i in range(nd):
        ref = torch.as_tensor(gop.infos[i]['ori']).cuda() + torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32)
        lossless = lossless and bool(torch.equal(dec[i], ref))
    torch.cuda.synchronize()
    t0 = time.time()
    codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', frames=list(range(nd)), workers=nd)
    torch.cuda.synchronize()
    decode_s = (time.time() - t0) / nd
    log('decode frames 0..%d: %.3f s/frame, lossless=%s' % (nd - 1, decode_s, lossless))
    decode_pts = {}
    for w in (1, 8):
        if w > len(gop):
            continue
        best = 1e9
        for rep in range(3):          # best of three: a shared host has bursts that last longer than one repetition
            shell = overfit.gen_model(gop.scale_num, 'cuda')
            tm = {}
            torch.cuda.synchronize()
            t0 = time.time()
            codec.decode_gop(shell, enc, 'cuda', frames=list(range(w)), workers=w, timing=tm)
            torch.cuda.synchronize()
            dt = time.time() - t0
            if (dt - tm['setup_s']) / w < best:
                best, decode_pts['gop_setup_s'] = (dt - tm['setup_s']) / w, tm['setup_s']
        decode_pts[w] = best
    log('decode: %s' % {k: round(v, 4) for k, v in decode_pts.items()})
    return lossless, decode_s, decode_pts, nd


def wide_leg(h):
    """--hidden_channel_conv 16 (main.py:520): the channel-blocked executor, a few training steps on frame 0.  Reported beside the
    headline (which is the reference's default width 8), never instead of it."""
    from linr_pcgc_amd import overfit
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    gop = h.gop
    try:
        mw = overfit.gen_model(gop.scale_num, 'cuda', seed=8807, hidden=16)
        ow = FlatAdam(mw)
        bw = torch.zeros(1, dtype=torch.float64, device='cuda')
        for _ in range(3):          # the first step builds the executor's buffer pool
            train_step(mw, ow, gop.frames[0], gop.point_nums[0], out=bw)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            train_step(mw, ow, gop.frames[0], gop.point_nums[0], out=bw)
        torch.cuda.synchronize()
        leg = {'hidden_channel_conv': 16, 'ms_per_step': round((time.time() - t0) * 1e3 / 10, 2), 'steps_timed': 10, 'parameters': int(mw.flat_parameters().numel()),
               'executor': 'channel-blocked (linr_pcgc_amd/wide_net.py) on csrc/wide.hip: a convolution, its backward-data, its weight gradient '
                           '(one gather per input block for all gradient blocks), a pointwise layer, a head and the backward of all 8 heads '
                           'are one launch each; the scale context runs on the 8-wide kernels; Python schedule'}
        del mw, ow
    except Exception as e:
        leg = {'error': repr(e)}
    log('hidden_channel_conv 16: %s' % leg)
    return leg


def bf16_codec_leg(h, enc, nd, barrier):
    """bf16 / uint8-weight codec leg (BASELINE config[4]'s numerics on this workload): the SAME trained model coded with the bf16
    executor (features bf16, weights as the uint8 codes of model.bin, de-quantised in-kernel).  Reported beside the fp32 headline,
    never instead of it.  Returns (leg, lossless)."""
    from linr_pcgc_amd import codec, overfit
    gop, model = h.gop, h.model
    try:
        from linr_pcgc_amd.model_codec import Model_Estimate
        barrier()
        t0 = time.time()
        enc_bf = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8, precision='bf16')
        barrier()
        bf_codec_s = time.time() - t0
        dec_bf = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc_bf, 'cuda', frames=list(range(nd)), workers=nd)
        bf_lossless = all(bool(torch.equal(dec_bf[i], torch.as_tensor(gop.infos[i]['ori']).cuda() +
                                           torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32))) for i in range(nd))
        coded = Model_Estimate().compress_model(model, 8, True, overfit.gen_model(gop.scale_num, 'cuda'))['new_model']
        fwd = {}
        for prec in ('f32', 'bf16'):
            fwd[prec] = _time_launches(lambda: coded.frame_probs(gop.frames[0], precision=prec), 20) * 1e3
        rows0 = gop.frames[0].rows
        # algorithmic bytes of one inference forward at 2-byte features: 48 conv3 x (2*(8+8) + 108) per row (SURVEY.md 8d form)
        leg = {'dtype': 'bf16', 'weights': 'uint8 codes of quant_uniform2, de-quantised in-kernel',
               'codec_s_per_frame': round(bf_codec_s / len(gop), 5), 'bits_per_point': round(float(enc_bf['bpp']['bpp_all']), 5),
               'point_bpp': round(enc_bf['bpp']['point_bpp'], 6), 'point_bpp_fp32': round(enc['bpp']['point_bpp'], 6),
               'lossless_decode_frames0to3': bf_lossless,
               'forward_ms_per_frame': {k: round(v, 4) for k, v in fwd.items()},
               'forward_alg_gbs': {'bf16': round(rows0 * 48 * (2 * 16 + 108) / (fwd['bf16'] * 1e-3) / 1e9, 1),
                                   'f32': round(rows0 * 48 * (4 * 16 + 108) / (fwd['f32'] * 1e-3) / 1e9, 1)}}
        log('bf16 leg: %s' % leg)
        return leg, bf_lossless
    except Exception as e:
        log('bf16 leg failed: %r' % (e,))
        return {'error': repr(e)}, True


def bpp_seeds_leg(h, enc):
    """bits/point of ONE run is only good to a few per cent: the 10-epoch overfit is run-to-run deterministic but chaotic in the
    rounding.  Two more complete overfits from other initialisation seeds (untimed) show the spread."""
    from linr_pcgc_amd import codec, overfit
    from linr_pcgc_amd.model_core import FlatAdam
    gop = h.gop
    vals = [float(enc['bpp']['bpp_all'])]
    seeds = [8807, 8808, 8809]
    for sd_ in seeds[1:]:
        m2 = overfit.gen_model(gop.scale_num, 'cuda', seed=sd_)
        overfit.overfit_gop(m2, FlatAdam(m2), gop, EPOCHS)
        vals.append(float(codec.encode_gop(m2, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)['bpp']['bpp_all']))
        del m2
    out = {'seeds': seeds, 'values': [round(v, 5) for v in vals], 'mean': round(sum(vals) / len(vals), 5),
           'min': round(min(vals), 5), 'max': round(max(vals), 5),
           'note': 'complete %d-epoch overfits of the same GOP from three initialisation seeds; `bits_per_point` is seed 8807' % EPOCHS}
    log('bits/point over seeds: %s' % out)
    return out


def device_report(world, dist, local):
    """Who ran: every rank's device (name, index, PCI bus id) and the collective backend - a SCALE record then shows N ranks on N
    different devices.  The only collectives are a start-up barrier and the MAX / SUM of times and bit counts (no data path)."""
    prop = torch.cuda.get_device_properties(local)
    mine = {'rank': int(os.environ.get('RANK', 0)), 'local_rank': local, 'device_index': torch.cuda.current_device(), 'device_name': prop.name,
            'pci_bus_id': getattr(prop, 'pci_bus_id', None), 'hbm_gib': round(prop.total_memory / 2.0 ** 30, 1)}
    ranks = [mine]
    if dist is not None:
        box = [None] * world
        dist.all_gather_object(box, mine)
        ranks = box
    return {'ranks': ranks, 'world_size': world,
            'backend': ('%s (RCCL)' % dist.get_backend() if dist.get_backend() == 'nccl' else dist.get_backend()) if dist is not None else None,
            'distinct_devices': len({(r['device_index'], r.get('pci_bus_id')) for r in ranks})}


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

    overfit_s_per_frame = h.full_overfit_s / len(gop)
    value = (overfit_s_per_frame + codec_s_per_frame) / world
    out = None
    if rank == 0:
        out = {'metric': 'encode_sec_per_frame', 'value': round(value, 5), 'unit': 's/frame', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(h.ms_per_step, 4),
               'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE config[%s] stand-in: synthetic %s (%d-bit sphere shell r~%d, %s voxel thick, %d points and %d '
                                      'parent rows in frame 0, %d scales), 1 GOP of %d frames per GPU, first_epoch=%d, '
                                      'lr 0.01 StepLR(32,0.992) Adam wd 1e-4, seed 8807'
                                      % ({'sphere8': '0', 'loot10': '1', 'andrew10': '3', 'owlii11': '4'}.get(args.config, '?'), args.config,
                                         synthetic.CONFIGS[args.config]['bitdepth'], synthetic.CONFIGS[args.config]['radius'],
                                         '%g' % (2 * synthetic.CONFIGS[args.config]['thickness']), gop.point_nums[0], gop.frames[0].rows,
                                         gop.scale_num, len(gop), EPOCHS),
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
               'components_s_per_frame': {'overfit': round(overfit_s_per_frame, 5), 'codec_modelcomp_fwd_ac_write': round(codec_s_per_frame, 5),
                                          'codec_first_call': round(codec_cold_s / len(gop), 5),
                                          'decode_s_per_frame_4_in_flight': round(decode_s, 4),
                                          'decode_s_single_frame': round(decode_pts.get(1, 0.0), 4),
                                          'decode_s_per_frame_8_in_flight': round(decode_pts.get(8, 0.0), 4),
                                          'decode_gop_setup_s': round(decode_pts.get('gop_setup_s', 0.0), 4)},
               'bf16_codec': bf16_leg,
               'bf16_train': bf16_train,
               'hidden16': wide,
               'epoch_loss_bpp': [round(x, 4) for x in h.losses], 'setup_s': round(h.setup_s, 1),
               'reference_logged': {'train_s_per_frame_epoch': 0.55, 'codec_s_per_frame': 0.43,
                                    'source': 'loot/info.log, loot/gop_32_62/*/result.json (RTX 3090, real loot)'}}
        out['roofline'] = None if os.environ.get('LINR_SKIP_ROOFLINE') else kernel_roofline(gop, h.live, table_prof, TABLE_STEPS, h.ms_per_step)
        log('roofline: %s' % out['roofline'])
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
                                   'linr_coords_sort_unique, linr_octree_level); kernel_map = neighbour search, compressed map, tiled copy, '
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
