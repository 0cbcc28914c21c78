"""Roofline bookkeeping of bench.py: kernel classes, live launch timing read-out, stored counter bytes (profiles/traffic*.json)."""
import ctypes   # noqa: F401
import json     # noqa: F401
import os
import sys      # noqa: F401
import time     # noqa: F401

import numpy as np   # noqa: F401
import torch

from .common import HBM_PEAK_GBS, ROOT, _popcount32

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
    (3, 'conv_bwd_wgrad_k<2> fused backward of conv0_0 8->4 (+ conv1_0 backward-data in the epilogue)', 2 * 156 + 48,
        ['voidconv_bwd_wgrad_k<2,'],
     'backward-data + weight gradient of conv0_0, backward-data + weight gradient of the 1x1 conv1_0: 4 layer passes per launch pass'),
    (4, 'cconv_mfma_k<8,8,fwd,head> prune conv + head MLP + sigmoid + BCE', 172 + 228, ['voidcconv_mfma_k<8,8,false,8,1>'],
     'conv3 + Linear(8,24) + Linear(24,1) + BCE: 4 layer passes per launch pass'),
    (5, 'cconv_mfma_k<8,4,fwd,pw> conv0_0 + conv1_0', 156 + 48, ['voidcconv_mfma_k<8,4,false,8,2>'],
        'conv3 8->4 + 1x1 8->4: 2 layer passes per launch pass'),
    (6, 'cconv_dual44_k<fwd> both 4->4 convs + conv1_2 + residual', 280 + 32, ['voidcconv_dual44_k<false>'],
     'two conv3 4->4 + 1x1 4->4: 3 layer passes per launch pass'),
    (7, 'occ_conv7_k first convs of the 7 outter blocks (one gather; 7 layer passes)', 156, ['occ_conv7_k'],
     'seven first convolutions from one gather: the 7 layer passes are counted as passes of this launch'),
    (8, 'head_bwd_k head MLP backward (data + weights)', 2 * 228, ['head_bwd_k'],
        'backward-data + weight gradients of both Linear layers: 4 layer passes per launch pass'),
    (9, 'occ_wgrad7_k weight gradients of the first convs of the 7 outter blocks (one gather; 7 layer passes)', 156, ['occ_wgrad7_k',
        'voidspconv_wgrad_t_k'],
     'seven weight gradients from one gather: the 7 layer passes are counted as passes of this launch'),
    (10, 'xtg_wgrad_k pointwise weight gradients', None, ['voidxtg_wgrad_k'], None),
    (11, 'sce_fwd_k / sce_bwd_k scale context', None, ['sce_fwd_k', 'sce_bwd_all_k'], None),
    (12, 'sum8_k, wgrad_reduce_k, sce_emb_grad, adam_k, bits finish', None, ['sum8_k', 'wgrad_reduce_k', 'sce_emb_grad_all_k', 'adam_k',
        'bce_bits_finish_k'], None),
    (13, 'stand-alone backward-data convolutions (schedules without the fused backward)', 172,
     ['voidcconv_mfma_k<8,8,true', 'voidcconv_mfma_k<4,8,true', 'voidcconv_dual44_k<true>'], None),
]


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
            roof['step'].update({'traffic': int(counter_total),
                'traffic_over_algorithmic': round(counter_total / (STEP_ALG_BYTES_PER_ROW * mean_rows), 4),
                                 'frac_counter': round(counter_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 'traffic_note': 'sum of the counter bytes of every kernel class of a step; ' + traffic_source(traffic_all)
                                                 + ('; no counters for: ' + ', '.join(counter_missing) if counter_missing else '')})
        roof['kernels_note'] = ('%d extra steps with EVERY launch bracketed by a HIP event pair, outside the timed region (state '
                                'saved and restored); sum %.1f us = %.3f of the un-instrumented step'
                                % (table_steps, covered, covered / (ms_per_step * 1e3)))
    return roof
