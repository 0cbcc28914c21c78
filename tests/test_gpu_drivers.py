"""GPU tests of the GOP / sequence drivers, the checkpoint policy, the decoder (threads, one call per scale, separate process,
committed golden stream) and the mirrored reference modules.  Tolerances: tests/gpu_common.py.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import network as onet          # noqa: E402,F401
from oracle import octree as ooct           # noqa: E402,F401
from oracle import ac as oac                # noqa: E402,F401
from gpu_common import _dev, _close, _model_and_oracle, _grads_close_per_tensor          # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_best_epoch_checkpoint_policy(pkg):
    """The reference writes model.pth only when the epoch's mean loss improves (main.py:413-426,440-451): the encoder codes with,
    and the GOPs >= 1 warm-start from, the BEST epoch.  With a learning rate far too large the later epochs are worse than an
    earlier one: overfit_gop(keep='best') must leave model and optimiser exactly as they were at the end of that epoch
    (re-run with that many epochs and keep='last': same bits), keep='last' must not."""
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam
    clouds = [synthetic.sphere_shell(7, 40 + t) for t in range(3)]

    def run(epochs, keep):
        gop = overfit.Gop(None, clouds, None, 64, 'cuda')
        model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
        opt = FlatAdam(model, lr=0.3, gamma=1.3, step_size=1)           # grows every frame: the overfit diverges
        info = {}
        losses = overfit.overfit_gop(model, opt, gop, epochs, keep=keep, info=info)
        return gop, model, opt, losses, info

    gop, model, opt, losses, info = run(8, 'best')
    k = int(np.argmin(losses))
    assert k < 7, 'the test needs an overfit whose last epoch is not the best: %s' % losses
    assert info['coded_epoch'] == k and info['coded_loss'] == losses[k]
    _, m_ref, o_ref, l_ref, _ = run(k + 1, 'last')
    assert l_ref == losses[:k + 1]
    assert torch.equal(model.flat_parameters(), m_ref.flat_parameters())
    assert torch.equal(opt.exp_avg, o_ref.exp_avg) and torch.equal(opt.exp_avg_sq, o_ref.exp_avg_sq)
    assert opt.t == o_ref.t == 3 * (k + 1) and opt.t_scale.tolist() == o_ref.t_scale.tolist()
    _, m_last, _, l_last, i_last = run(8, 'last')
    assert l_last == losses and i_last['coded_epoch'] == 7
    assert not torch.equal(m_last.flat_parameters(), model.flat_parameters())
    # the kept model is the one that gets coded: fewer bits than the last epoch's
    enc_best = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    enc_last = codec.encode_gop(m_last, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    assert enc_best['bpp']['point_bpp'] < enc_last['bpp']['point_bpp']


def test_diverged_overfit_is_reported(pkg):
    """A learning rate that sends the parameters to NaN.  The loss alone does not show it (BCELoss's clamp turns a NaN probability
    into 100 nats, so the numbers stay finite): overfit_gop counts an epoch that ends with non-finite parameters as diverged
    (loss = inf), keep='best' hands back the last sound epoch when there is one, and an overfit without one raises instead of
    passing NaN parameters on to the model codec (where the reference's quantiser assertion, model_size_est.py:81, would be the
    first thing to notice)."""
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam
    gop = overfit.Gop(None, [synthetic.sphere_shell(7, 40), synthetic.sphere_shell(7, 41)], None, 64, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    with pytest.raises(FloatingPointError, match='diverged'):
        overfit.overfit_gop(model, FlatAdam(model, lr=1e8), gop, 3, keep='last')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    with pytest.raises(FloatingPointError, match='diverged'):
        overfit.overfit_gop(model, FlatAdam(model, lr=1e8), gop, 3, keep='best')
    # sound for two epochs, then the learning rate explodes: the second epoch is kept and can be coded
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    opt = FlatAdam(model, lr=0.01)
    info = {}

    def blow_up(epoch, loss):
        if epoch == 1:
            opt.lr = 1e8
    losses = overfit.overfit_gop(model, opt, gop, 4, keep='best', info=info, on_epoch=blow_up)
    assert math.isfinite(losses[0]) and math.isfinite(losses[1]) and losses[2] == float('inf') and losses[3] == float('inf')
    assert info['coded_epoch'] == 1 and bool(torch.isfinite(model.flat_parameters()).all())
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    assert math.isfinite(enc['bpp']['bpp_all'])


def test_concurrent_training_in_threads_equals_serial(pkg):
    """The training path is re-entrant too (include/linr_hip.h: no mutable global state on the data path): three GOPs overfitted
    and encoded at the same time by three host threads on three streams end in the losses, parameters and stream bytes of the
    same three jobs run one after the other.  (Models are built beforehand: torch's initialisation draws from a process-wide RNG.)"""
    import threading
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam

    def make(k):
        gop = overfit.Gop(None, [synthetic.sphere_shell(7, 30 + 5 * k + t) for t in range(3)], None, 64, 'cuda')
        return gop, overfit.gen_model(gop.scale_num, 'cuda', seed=100 + k), overfit.gen_model(gop.scale_num, 'cuda')

    def job(gop, model, shell, out, stream):
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            losses = overfit.overfit_gop(model, FlatAdam(model), gop, 4)
            enc = codec.encode_gop(model, shell, gop, 8, n_threads=2)
            stream.synchronize()
        out.append((losses, [bytes(b) for f in enc['frames'] for b in f], enc['model_bin'], model.flat_parameters().clone()))
    serial = []
    for k in range(3):
        job(*make(k), serial, torch.cuda.current_stream())
    sets = [make(k) for k in range(3)]
    torch.cuda.synchronize()
    res = [[] for _ in range(3)]
    threads = [threading.Thread(target=job, args=(*sets[k], res[k], torch.cuda.Stream())) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(3):
        assert len(res[k]) == 1, 'thread %d died' % k
        a, b = serial[k], res[k][0]
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and torch.equal(a[3], b[3]), k


def test_threaded_gop_decode_equals_serial(pkg):
    """codec.decode_gop(workers=3): frames decoded concurrently on their own streams give the serial result."""
    from linr_pcgc_amd import codec, overfit, synthetic
    clouds = [synthetic.sphere_shell(7, 40 + t) for t in range(5)]
    gop = overfit.Gop(None, clouds, None, 64, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    serial = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda')
    threaded = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda', workers=3)
    for i, (a, b) in enumerate(zip(serial, threaded)):
        assert torch.equal(a, b), i
        ref = torch.as_tensor(gop.infos[i]['ori']).cuda() + torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32)
        assert torch.equal(a, ref)


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
def test_decode_scale_call_equals_stagewise_decode(pkg, precision):
    """linr_decode_scale (kernel map + 8 decode stages + upper_layer of a scale in one C call) against the reference-shaped
    path (model.decode per scale, octree_level.upper_layer in torch): the same coordinates, level by level; a child buffer that
    is too small is refused."""
    import ctypes
    from linr_pcgc_amd import _lib, codec, overfit, synthetic
    from linr_pcgc_amd.model_codec import Model_Estimate
    from linr_pcgc_amd.module_utils import unique_sorted
    clouds = [synthetic.sphere_shell(7, 41), synthetic.sphere_shell(7, 47)]
    gop = overfit.Gop(None, clouds, None, 64, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8, precision=precision)
    side = dict(enc['side_info'])
    side.pop('arith_version', None)
    side['final_bytes'] = enc['model_bin']
    m, _ = Model_Estimate().decompress_model(overfit.gen_model(gop.scale_num, 'cuda'), side)
    m.inference_precision = precision
    lows, mins = codec.dec_all_frame_low_xyz(enc['low_enc_bytes'])
    for i in range(2):
        xyz_low = torch.tensor(lows[i].astype(np.int32), device='cuda')
        a = codec.decode_one_frame(m, list(enc['frames'][i]), xyz_low)['dec_coord']
        b = codec.decode_one_frame_stagewise(m, list(enc['frames'][i]), xyz_low)['dec_coord']
        assert torch.equal(a, b)
        ref = torch.as_tensor(gop.infos[i]['ori']).cuda()
        assert torch.equal(a, ref)
    # argument checks of the entry: capacity of the child buffer, alignment of the workspace
    L = _lib.lib()
    lowx = unique_sorted(torch.tensor(lows[0].astype(np.int32), device='cuda')).contiguous()
    n = lowx.shape[0]
    from linr_pcgc_amd.function_utils import unpack_bitstream
    streams = [np.frombuffer(b, dtype=np.uint8) for b in unpack_bitstream(enc['frames'][0][-1])]
    ptrs = (ctypes.c_void_p * 8)(*[b.ctypes.data if b.size else None for b in streams])
    lens = (ctypes.c_int64 * 8)(*[int(b.size) for b in streams])
    need = L.linr_decode_scale_ws_bytes(n, 1, 1 if precision == 'bf16' else 0)
    ws = torch.empty(need + 512, dtype=torch.uint8, device='cuda')
    base = (ws.data_ptr() + 255) & ~255
    p_host, s_host = m._host_buffers(n)
    child = torch.empty((8 * n, 3), dtype=torch.int32, device='cuda')
    cnt = ctypes.c_int64(0)
    params = None if precision == 'bf16' else m.flat_parameters().data_ptr()
    codes = m._qcodes.data_ptr() if precision == 'bf16' else None
    lo, hi = (float(m._qrange[0]), float(m._qrange[1])) if precision == 'bf16' else (0.0, 0.0)
    args = lambda ws_ptr, cap: (lowx.data_ptr(), n, gop.scale_num - 1, gop.scale_num, 1, 8, params, codes, lo, hi, ptrs, lens, ws_ptr, need,
                                p_host.data_ptr(), s_host.data_ptr(), child.data_ptr(), cap, ctypes.byref(cnt),
                                torch.cuda.current_stream().cuda_stream)
    assert L.linr_decode_scale(*args(base + 8, 8 * n)) == -3
    assert L.linr_decode_scale(*args(base, 1)) == -2
    assert L.linr_decode_scale(*args(base, 8 * n)) == 0 and 0 < cnt.value <= 8 * n


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
def test_committed_stream_still_decodes(pkg, golden_dir, precision):
    """The fp32 evaluation order of the forward is part of the stream format (codec.ARITH_VERSION).  A stream coded by the build
    that set the current version is committed (tests/golden/stream_v*.npz, written by tests/golden/make_stream_golden.py on a
    MI355X): this build must decode it to the same geometry - a change of the arithmetic without a version bump fails here."""
    import ast
    from linr_pcgc_amd import codec, overfit
    path = os.path.join(golden_dir, 'stream_v%d.npz' % codec.ARITH_VERSION)
    assert os.path.exists(path), 'no committed stream for ARITH_VERSION %d: run tests/golden/make_stream_golden.py' % codec.ARITH_VERSION
    g = np.load(path)
    assert int(g['arith_version']) == codec.ARITH_VERSION
    n_scales = int(g['scale_num'])
    frames = [[g['%s_f%d_s%d' % (precision, fi, si)].tobytes() for si in range(n_scales)] for fi in range(2)]
    enc = {'frames': frames, 'model_bin': g[precision + '_model_bin'].tobytes(), 'low_enc_bytes': g[precision + '_low'].tobytes(),
           'side_info': ast.literal_eval(str(g[precision + '_side']))}
    dec = codec.decode_gop(overfit.gen_model(n_scales, 'cuda'), enc, 'cuda')
    for i in range(2):
        assert torch.equal(dec[i].cpu(), torch.from_numpy(g['ref%d' % i])), 'frame %d of the committed stream decodes to other geometry' % i


def test_gop_flow_checkpoint_warm_start_files(pkg, tmp_path):
    """main.overfit_enc_dec in miniature: GOP 0 from scratch -> checkpoint -> GOP 1 warm start (model + Adam state) ->
    encode -> reference directory layout on disk -> decode from the files alone -> lossless; model codec round trip."""
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_codec import Model_Estimate
    from linr_pcgc_amd.model_core import FlatAdam
    clouds = [synthetic.sphere_shell(6, 20 + t) for t in range(4)]
    gop0 = overfit.Gop(None, clouds[:2], None, 64, 'cuda')
    model = overfit.gen_model(gop0.scale_num, 'cuda', seed=8807)
    opt = FlatAdam(model)
    l0 = overfit.overfit_gop(model, opt, gop0, 3)
    assert l0[-1] < l0[0]
    ck = overfit.checkpoint(model, opt, 2, l0[-1])
    assert set(ck) == {'model', 'epoch', 'optimizer_state_dict', 'loss', 'bitdepth'}
    assert len(ck['model']) == 1 + 4 * gop0.scale_num + 160           # 189 tensors at scale_num 7
    torch.save(ck, tmp_path / 'model.pth')
    ck2 = torch.load(tmp_path / 'model.pth', weights_only=False)
    gop1 = overfit.Gop(None, clouds[2:], gop0.scale_num, 64, 'cuda')
    m1 = overfit.gen_model(gop0.scale_num, 'cuda', seed=1)
    o1 = FlatAdam(m1)
    overfit.warm_start(m1, o1, ck2)
    assert torch.equal(m1.flat_parameters(), model.flat_parameters()) and o1.t == opt.t and abs(o1.lr - opt.lr) < 1e-15
    assert torch.equal(o1.exp_avg, opt.exp_avg)
    l1 = overfit.overfit_gop(m1, o1, gop1, 2)
    assert l1[0] < l0[0]                                         # warm start begins far below the cold-start loss
    est = Model_Estimate()
    test = est.compress_test(m1, overfit.gen_model(gop0.scale_num, 'cuda'), 8)
    assert test['enc_mode'] in (0, 1, 2) and test['bit_real'] > 0
    enc = codec.encode_gop(m1, overfit.gen_model(gop0.scale_num, 'cuda'), gop1, 8)
    codec.write_gop(enc, str(tmp_path / 'gop_2_3'))
    names = sorted(os.listdir(tmp_path / 'gop_2_3' / 'bins'))
    assert 'model.bin' in names and 'low_enc_bytes.bin' in names and 'frame0000_scale0.bin' in names
    back = codec.read_gop(str(tmp_path / 'gop_2_3'))
    assert back['frames'] == enc['frames'] and back['model_bin'] == enc['model_bin']
    dec = codec.decode_gop(overfit.gen_model(gop0.scale_num, 'cuda'), back, 'cuda')
    for d, info, mn in zip(dec, gop1.infos, gop1.coord_mins):
        ref = torch.as_tensor(info['ori']).cuda() + torch.tensor(mn, device='cuda', dtype=torch.int32)
        assert torch.equal(d, ref)
    assert 0 < enc['bpp']['point_bpp'] < 8 and enc['bpp']['model_bpp'] > 0      # tiny clouds: the 35 KB model dominates bpp_all


def test_config2_sequence_300_frames_gop32(pkg, tmp_path):
    """BASELINE config[2] on one GPU: the 300-frame loot10 stand-in in GOPs of 32 (10 GOPs, the last one 12 frames),
    first_epoch = others_epoch = 1: GOP 0 from scratch, GOPs 1..9 warm-started from its checkpoint file (model + Adam
    state), every GOP encoded to the reference's directory layout and EVERY frame decoded from the files and compared
    bit for bit (main.py:83-104, encoder.py:57-156, decoder.py:51-146)."""
    from linr_pcgc_amd import gop_parallel, run
    out = str(tmp_path / 'seq')
    args = run.parse(['--config', 'loot10', '--frames', '300', '--gop', '32', '--first-epoch', '1', '--others-epoch', '1',
                      '--out', out, '--decode', '--schedule', 'pull'])
    summary, results = run.run_sequence_job(args, 0, 1, None)
    assert summary['gops'] == 10 and summary['frames'] == 300 and summary['lossless'] is True
    assert sorted(results) == list(range(10))
    assert [results[g]['frames'] for g in range(10)] == [32] * 9 + [12]
    assert all(r['lossless'] for r in results.values())
    assert abs(summary['ideal_speedup_bound'] - 1.0) < 1e-9                  # one GPU
    assert abs(gop_parallel.ideal_speedup(gop_parallel.split_gops(300, 32), 8) - 300 / 76.0) < 1e-9
    # warm start: after ONE epoch every later GOP is already far below GOP 0's from-scratch loss
    assert all(results[g]['loss'][-1] < 0.8 * results[0]['loss'][-1] for g in range(1, 10))
    assert 0.2 < summary['bits_per_point'] < 3.0
    assert os.path.exists(os.path.join(out, 'output', 'gop_0_31', 'model.pth'))
    assert os.path.exists(os.path.join(out, 'result_enc', 'gop_288_299', 'bins', 'frame0011_scale0.bin'))


def test_mid_test_driver_writes_the_reference_result_files(pkg, tmp_path):
    """test_utils.Test_one_gop (test_utils.py:16-163; main.py:365-380 calls it on a checkpoint): the reference's argument dict in,
    result.json / side_info.json / bins out, the numbers consistent with the GOP encoder's (same model codec, same low-resolution
    payload; one stream per scale instead of eight, so the occupancy rate agrees to the coder's termination overhead)."""
    import json
    from linr_pcgc_amd import codec, overfit, synthetic, test_utils
    from linr_pcgc_amd.model_codec import Model_Estimate
    from linr_pcgc_amd.model_core import FlatAdam
    clouds = [synthetic.sphere_shell(7, 40 + t) for t in range(2)]
    gop = overfit.Gop(None, clouds, None, 64, 'cuda')
    gen = lambda: overfit.gen_model(gop.scale_num, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    opt = FlatAdam(model)
    overfit.overfit_gop(model, opt, gop, 3)
    ck_path = str(tmp_path / 'model.pth')
    torch.save(overfit.checkpoint(model, opt, 2, 0.0), ck_path)
    low = codec.enc_all_frame_low_xyz(gop)
    reading = [{'all_input_info': fr['all_input_info'], 'point_num': fr['point_num']} for fr in gop.infos]
    args = {'model_path': ck_path, 'Gen_Model': gen, 'frame_num': 2, 'compress_model_test': Model_Estimate().compress_test,
            'reading_data': reading, 'result_dir': str(tmp_path / '2'), 'write_flag': True, 'low_enc_ret': low}
    res = test_utils.Test_one_gop(args)
    assert sorted(res) == ['bpp_all', 'dec_time', 'enc_mode', 'enc_time', 'model_bpp', 'point_bpp', 'point_bpp_val', 'xyzlow_bpp']
    assert res == json.load(open(str(tmp_path / '2' / 'result.json')))
    side = json.load(open(str(tmp_path / '2' / 'side_info.json')))
    assert sorted(side) == ['b', 'enc_mode', 'max_param', 'min_param', 'mu', 'xlow_enc_flags', 'xlow_enc_modes']
    for name in ('model.bin', 'low_enc_bytes.bin', 'frame0000_scale0.bin', 'frame0001_scale%d.bin' % (gop.scale_num - 1)):
        assert os.path.getsize(str(tmp_path / '2' / 'bins' / name)) > 0
    enc = codec.encode_gop(model, gen(), gop, 8)
    points = sum(gop.point_nums)          # the GOP encoder also counts its extra side-info bytes (arith_version, precision, model shape)
    assert abs(res['model_bpp'] + codec.EXTRA_SIDE_BITS / points - enc['bpp']['model_bpp']) < 1e-9
    assert abs(res['xyzlow_bpp'] - enc['bpp']['xyzlow_bpp']) < 1e-12
    # 48 streams with their length fields instead of 6 per frame: a few per cent on clouds this small
    assert res['point_bpp'] <= enc['bpp']['point_bpp'] <= 1.05 * res['point_bpp']
    assert abs(res['point_bpp_val'] - res['point_bpp']) <= 0.02 * res['point_bpp']          # coded size tracks the loss
    assert abs(res['bpp_all'] - (res['point_bpp'] + res['model_bpp'] + res['xyzlow_bpp'])) < 1e-12
    assert res['enc_time'] > 0 and res['dec_time'] > 0
    with pytest.raises(ValueError):
        test_utils.Test_one_gop(dict(args, low_enc_ret=None))
    res2 = test_utils.Test_one_gop(dict(args, write_flag=False, result_dir=str(tmp_path / 'nowrite')))
    assert res2['bpp_all'] == res['bpp_all'] and not os.path.exists(str(tmp_path / 'nowrite' / 'bins' / 'model.bin'))


def test_run_with_mid_test_matches_the_plain_run(pkg, tmp_path):
    """run.py --mid-test (main.py --mid_test, :341-411): Test_one_gop at every epoch < 10 into <out>/output/<gop>/<epoch>/ and the
    per-epoch list in <gop>/result.json - and the training it observes is bit for bit the training of a run without it."""
    import json
    from linr_pcgc_amd import run
    base = ['--config', 'sphere8', '--frames', '4', '--gop', '2', '--first-epoch', '3', '--others-epoch', '2']
    s0, r0 = run.run_sequence_job(run.parse(base + ['--out', str(tmp_path / 'plain')]), 0, 1, None)
    s1, r1 = run.run_sequence_job(run.parse(base + ['--out', str(tmp_path / 'mid'), '--mid-test']), 0, 1, None)
    assert [r0[g]['loss'] for g in (0, 1)] == [r1[g]['loss'] for g in (0, 1)] and s0['bits_per_point'] == s1['bits_per_point']
    for g, name, epochs in ((0, 'gop_0_1', 3), (1, 'gop_2_3', 2)):
        lst = json.load(open(str(tmp_path / 'mid' / 'output' / name / 'result.json')))
        assert [e['epoch'] for e in lst] == list(range(epochs)) and lst == r1[g]['mid_test']
        for e in lst:
            one = json.load(open(str(tmp_path / 'mid' / 'output' / name / str(e['epoch']) / 'result.json')))
            assert one['bpp_all'] == e['real_bpp_all']
            # the loss of the epoch is the mean over its steps, the mid-test sees the state behind the last one (quantised)
            assert e['point_bpp_val'] < e['loss'] and abs(e['real_point_bpp'] - e['point_bpp_val']) <= 0.03 * e['point_bpp_val']
            assert not os.path.exists(str(tmp_path / 'mid' / 'output' / name / str(e['epoch']) / 'bins' / 'model.bin'))
        assert lst[-1]['real_bpp_all'] < lst[0]['real_bpp_all']          # it is learning, and the mid-test sees it
    assert 'mid_test' not in r0[0]
    # info.log and the per-epoch list exist without --mid-test too, in the reference's format (main.py:327-338,428-430)
    plain = json.load(open(str(tmp_path / 'plain' / 'output' / 'gop_0_1' / 'result.json')))
    assert [sorted(e) for e in plain] == [['epoch', 'loss', 'train_time', 'train_time_avg']] * 3 and [e['loss'] for e in plain] == r0[0]['loss']
    assert plain[2]['train_time'] > plain[0]['train_time'] > 0 and abs(plain[1]['train_time_avg'] - plain[1]['train_time'] / 2) < 1e-12
    lines = open(str(tmp_path / 'plain' / 'info.log')).read().splitlines()
    assert lines[0] == '=' * 40 and lines[1] == 'process_file: 0 1' and lines[2] == 'epoch: 0' and lines[3].startswith('loss: ')
    assert 'process_file: 2 3' in lines and sum(1 for ln in lines if ln.startswith('epoch: ')) == 5


def test_reference_driver_flow_on_the_mirrored_modules(pkg, tmp_path):
    """The calls the reference's own drivers make, in their order, on this package's modules: MyDataset / Read_Data (main.py:73-78,
    147), overfit_one_frame's loop with model(putin_args), loss.backward() and torch.optim.Adam (main.py:305-321,457-475), the
    estimate-vs-codec check (main.py:290-295), Test_one_gop on the checkpoint (main.py:377), encode_one_frame's model.encode per
    scale (encoder.py:158-176), decode_one_frame (decoder.py:153-176) and the comparison with MytestDataset (decoder.py:118-131)."""
    from linr_pcgc_amd import codec, custom_dataset as cd, synthetic, test_utils
    from linr_pcgc_amd.model_codec import Model_Estimate
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    ori = tmp_path / 'ori'
    ori.mkdir()
    for t in range(2):
        np.save(str(ori / ('frame_%04d.npy' % t)), synthetic.sphere_shell(7, 40 + t))
        cd.write_ply_ascii(str(ori / ('frame_%04d.ply' % t)), synthetic.sphere_shell(7, 40 + t))
    dataset = cd.MyDataset(str(ori), str(tmp_path / 'handle'), None, 'npy', stage=8)
    dataset.set_prefix_data({'offsets_ini': torch.tensor(cd.OFFSETS_INI, device='cuda'), 'min_point_num': 64})
    dataset[0]
    gen = lambda: LINR_PCGC_Model({'scale_num': dataset.scale_num, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1,
                                   'outstage': 8, 'instage': 1}).cuda()
    reading = cd.Read_Data(dataset, [0, 1])
    torch.manual_seed(8807)
    model = gen()
    est, real = Model_Estimate().estibits(model, gen(), 8), Model_Estimate().compress_test(model, gen(), 8)
    assert int((est['recon_ret'] != real['recon_ret']).sum()) == 0
    optim = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-4)

    def putin(inargs):
        d = dict(inargs)
        d['coord'], d['offset_tensor'] = inargs['xyzqsc_t'].get_coord(), inargs['xyzqsc_t'].get_offset_tensor()
        return d
    epoch_loss = []
    for epoch in range(3):
        tot = 0.0
        for fi in range(len(reading)):
            frame = reading[fi]
            bits = 0
            for inargs in frame['all_input_info']:
                bits = bits + model(putin(inargs))
            loss = bits / frame['point_num']
            optim.zero_grad()
            loss.backward()
            optim.step()
            tot += float(loss.detach())
        epoch_loss.append(tot / len(reading))
    assert epoch_loss[-1] < epoch_loss[0]
    ck = str(tmp_path / 'model.pth')
    torch.save({'model': model.state_dict(), 'epoch': 2, 'optimizer_state_dict': optim.state_dict(), 'loss': epoch_loss[-1], 'bitdepth': 8}, ck)
    low = test_utils.enc_all_frame_low_xyz(reading, 2)
    assert low == _low_xyz_bytes(reading)
    res = test_utils.Test_one_gop({'model_path': ck, 'Gen_Model': gen, 'frame_num': 2, 'compress_model_test': Model_Estimate().compress_test,
                                   'reading_data': reading, 'result_dir': str(tmp_path / '2'), 'write_flag': False, 'low_enc_ret': low})
    assert 0 < res['point_bpp'] < 1.1 * epoch_loss[-1] + 0.5
    # encoder.encode / decoder.decode with the reference's argument dicts (main.py:110-115): files out, frames back from the files
    # alone, compared with the test data set's sorted voxel lists and written as PLY
    from linr_pcgc_amd import decoder, encoder
    out_dir = tmp_path / 'output' / 'gop_0_1'
    out_dir.mkdir(parents=True)
    os.replace(ck, str(out_dir / 'model.pth'))
    encoder.encode({'outputdir': str(tmp_path / 'output'), 'gop_names': ['gop_0_1'], 'Gen_Model': gen, 'dataset': dataset,
                    'encode_dir': str(tmp_path / 'enc')})
    for name in ('side_info.json', 'bins/model.bin', 'bins/low_enc_bytes.bin', 'bins/frame0001_scale0.bin'):
        assert os.path.getsize(str(tmp_path / 'enc' / 'gop_0_1' / name)) > 0
    test_set = cd.MytestDataset(str(ori), ori_type='ply')
    decoder.decode({'gop_names': ['gop_0_1'], 'Gen_Model': gen, 'result_enc_dir': str(tmp_path / 'enc'),
                    'result_dec_dir': str(tmp_path / 'dec'), 'dataset': test_set, 'write_flag': True})
    for fi in range(2):
        back = cd.read_ply_o3d(str(tmp_path / 'dec' / ('frame%04d.ply' % fi)))
        assert np.array_equal(back, synthetic.sphere_shell(7, 40 + fi))
    low_dec = test_utils.dec_all_frame_low_xyz(low)
    assert len(low_dec['all_xyz_low']) == 2 and low_dec['all_coord_data_min'].shape == (2, 3)
    # a corrupted stream must be noticed by the decoder's comparison
    path = str(tmp_path / 'enc' / 'gop_0_1' / 'bins' / 'frame0000_scale0.bin')
    blob = bytearray(open(path, 'rb').read())
    blob[len(blob) // 2] ^= 0x55
    open(path, 'wb').write(bytes(blob))
    with pytest.raises(AssertionError):
        decoder.decode({'gop_names': ['gop_0_1'], 'Gen_Model': gen, 'result_enc_dir': str(tmp_path / 'enc'),
                        'result_dec_dir': str(tmp_path / 'dec2'), 'dataset': test_set, 'write_flag': False})


def _low_xyz_bytes(reading):
    """test_utils.enc_all_frame_low_xyz (test_utils.py:199-232) on a Read_Data window: uint8 coarsest coordinates + int32 minima."""
    from linr_pcgc_amd.function_utils import pack_bitstream
    chunks = [reading[i]['all_input_info'][-1]['xyzqsc_t'].get_coord().cpu().numpy().astype(np.uint8).tobytes() for i in range(len(reading))]
    chunks.append(np.asarray([reading[i]['coord_data_min'] for i in range(len(reading))], dtype=np.int32).reshape(-1).tobytes())
    return pack_bitstream(chunks)


def test_decoder_as_a_separate_process(pkg, tmp_path):
    """Encoder and decoder are different programs in practice: the streams written by this process are decoded by a fresh
    interpreter (python -m linr_pcgc_amd.decoder: its own HIP context, other addresses, the library loaded anew) from the files
    alone, compared there with the input files and written as PLY."""
    import subprocess
    from linr_pcgc_amd import custom_dataset as cd, ply, run, synthetic
    ori = tmp_path / 'ori'
    ori.mkdir()
    files = []
    for t in range(3):
        path = str(ori / ('frame_%04d.ply' % t))
        ply.write_ply_xyz(path, synthetic.sphere_shell(7, 39 + t, centre=(60 + t, 64, 66)), binary=False)
        files.append(path)
    out = str(tmp_path / 'seq')
    args = run.parse(['--input-glob', str(ori / 'frame_*.ply'), '--frames', '3', '--gop', '2', '--first-epoch', '2', '--others-epoch', '1', '--out', out])
    summary, _ = run.run_sequence_job(args, 0, 1, None, files=files)
    assert summary['gops'] == 2 and summary['lossless'] is None          # nothing was decoded in this process
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    done = subprocess.run([sys.executable, '-m', 'linr_pcgc_amd.decoder', '--enc-dir', os.path.join(out, 'result_enc'), '--dec-dir',
                           str(tmp_path / 'dec'), '--ori-dir', str(ori)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    assert 'decoded 3 frames of 2 GOPs' in done.stdout and 'all equal to the input' in done.stdout
    for t in range(3):
        assert np.array_equal(cd.read_ply_o3d(str(tmp_path / 'dec' / ('frame%04d.ply' % t))), synthetic.sphere_shell(7, 39 + t, centre=(60 + t, 64, 66)))


def _run_npy_sequence(tmp_path, tag, clouds, extra=()):
    from linr_pcgc_amd import run
    ori = tmp_path / (tag + '_ori')
    ori.mkdir()
    for i, c in enumerate(clouds):
        np.save(str(ori / ('f%03d.npy' % i)), np.asarray(c))
    args = run.parse(['--ori_dir', str(ori), '--ori_dtype', 'npy', '--frame_num', str(len(clouds)), '--gop_size', '2', '--first_epoch', '2',
                      '--others_epoch', '1', '--result_dir', str(tmp_path / (tag + '_out')), '--decode'] + list(extra))
    return run.run_sequence_job(args, 0, 1, None, files=run.resolve_files(args))


def test_sequence_scale_count_is_fixed_by_the_first_frame(pkg, tmp_path):
    """main.py:73-78: dataset[0] fixes scale_num for the WHOLE sequence; GOPs >= 1 load GOP 0's checkpoint, so their models must
    have its shape whatever their own frames look like.  Here the frames of later GOPs would have more scales (growing clouds) or
    fewer (a 3-voxel-radius blob leads a GOP) than frame 0 - the first used to build a larger model, the second a smaller one, and
    the warm start failed.  Also: negative coordinates, and a frame list whose sizes differ by two orders of magnitude."""
    from linr_pcgc_amd import synthetic
    sph = lambda r: synthetic.sphere_shell(7, r)
    s, r = _run_npy_sequence(tmp_path, 'grow', [sph(30) - 200, sph(31) - 200, sph(32) - 200])
    assert s['lossless'] is True and sorted(r) == [0, 1]
    s, r = _run_npy_sequence(tmp_path, 'mixed', [sph(40), sph(10), sph(3), sph(45)])
    assert s['lossless'] is True and sorted(r) == [0, 1]
    import json
    side = json.load(open(str(tmp_path / 'mixed_out' / 'result_enc' / 'gop_2_3' / 'side_info.json')))
    first = json.load(open(str(tmp_path / 'mixed_out' / 'result_enc' / 'gop_0_1' / 'side_info.json')))
    assert side['scale_num'] == first['scale_num'] and side['block_layers'] == 1 and side['hidden_channel_conv'] == 8
    # GOP 2..3 starts with the 3-voxel blob: its frame 0 has fewer scale streams than the model has scales
    n_streams = len([f for f in os.listdir(str(tmp_path / 'mixed_out' / 'result_enc' / 'gop_2_3' / 'bins')) if f.startswith('frame0000_scale')])
    assert n_streams < side['scale_num']


def test_sequence_degenerate_clouds(pkg, tmp_path):
    """A single point, two points, a few dozen scattered points, repeated points, float coordinates, a plane and a line: every one
    goes through the whole flow losslessly (the rates are absurd - the model costs more than the points - but nothing breaks); a
    cloud wider than the 20-bit coordinates of the kernel map is refused with a message that says so."""
    from linr_pcgc_amd import synthetic
    rng = np.random.default_rng(0)
    sph = lambda r: synthetic.sphere_shell(7, r)
    cases = {'tiny': [np.array([[5, 6, 7]]), np.array([[1, 2, 3], [1, 2, 4]])],
             'few': [rng.integers(0, 64, size=(40, 3)), rng.integers(0, 64, size=(100, 3))],
             'dups': [np.repeat(sph(20), 3, axis=0), sph(21)],
             'floats': [sph(20).astype(np.float64) + 0.2, sph(21).astype(np.float32) - 0.3],
             'flat': [np.stack([rng.integers(0, 128, 5000), rng.integers(0, 128, 5000), np.zeros(5000, np.int64)], 1), sph(21)],
             'line': [np.stack([np.arange(300), np.zeros(300, np.int64), np.zeros(300, np.int64)], 1), sph(21)]}
    for tag, clouds in cases.items():
        s, _ = _run_npy_sequence(tmp_path, tag, clouds)
        assert s['lossless'] is True, tag
    with pytest.raises(ValueError, match='20-bit'):
        _run_npy_sequence(tmp_path, 'wide', [sph(30) * 20000])


def test_model_surface_rejects_malformed_inputs(pkg):
    """model(putin_args) / encode / decode with the per-scale dicts user code builds (main.py:457-475): dtype and device conversions
    are accepted, everything that cannot be right is refused with a message naming the field - including inputs that alias a
    cached frame's tensors (slices share data pointers) - and a scale without voxels costs zero bits."""
    from linr_pcgc_amd import overfit, synthetic
    from linr_pcgc_amd.module_utils import prepare_frame
    fr = prepare_frame(synthetic.sphere_shell(7, 30), None, 64, device='cuda')
    model = overfit.gen_model(fr['scale_num'], 'cuda', seed=1)
    s = fr['all_input_info'][0]
    n = int(s['coord'].shape[0])

    def putin(**over):
        d = {'coord': s['coord'], 'offset_tensor': s['offset_tensor'], 'occ_lst': s['occ_lst'], 'scale_idx': 0}
        d.update(over)
        return d
    with torch.no_grad():
        base = float(model(putin()))
        same = [putin(coord=s['coord'].long()), putin(offset_tensor=None), putin(occ_lst=[o.double() for o in s['occ_lst']]),
                putin(occ_lst=[o.bool() for o in s['occ_lst']]),
                putin(coord=s['coord'].cpu(), offset_tensor=s['offset_tensor'].cpu(), occ_lst=[o.cpu() for o in s['occ_lst']])]
        assert all(float(model(d)) == base for d in same)
        perm = torch.randperm(n, device='cuda')
        bad = [(putin(coord=s['coord'][perm]), 'sorted'), (putin(coord=torch.cat([s['coord'][:1], s['coord'][:-1]])), 'unique'),
               (putin(coord=s['coord'] - 5), 'non-negative'), (putin(scale_idx=fr['scale_num']), 'scale_idx'), (putin(scale_idx=-1), 'scale_idx'),
               (putin(offset_tensor=s['offset_tensor'][:-1]), 'offset_tensor'), (putin(occ_lst=s['occ_lst'][:7]), 'occ_lst'),
               (putin(occ_lst=[o[:-1] for o in s['occ_lst']]), 'occ_lst'), (putin(coord=s['coord'][:, :2]), 'coord')]
        for d, word in bad:
            with pytest.raises(ValueError, match=word):
                model(d)
        assert float(model(putin(coord=s['coord'][:0], offset_tensor=s['offset_tensor'][:0], occ_lst=[o[:0] for o in s['occ_lst']]))) == 0.0
        enc = model.encode(putin())['enc_bytes']
        dec = model.decode({'enc_bytes': enc, 'coord': s['coord'], 'offset_tensor': None, 'scale_idx': 0})
        assert torch.equal(torch.cat(dec, dim=1), torch.cat([o.reshape(-1, 1) for o in s['occ_lst']], dim=1).float())
        for blob in (b'', enc[:len(enc) // 2], bytes(np.random.default_rng(0).integers(0, 256, 500, dtype=np.uint8))):
            with pytest.raises(ValueError, match='container'):
                model.decode({'enc_bytes': blob, 'coord': s['coord'], 'offset_tensor': None, 'scale_idx': 0})


def test_sequence_from_ply_files(pkg, tmp_path):
    """The driver on a real file sequence (main.py:69-119 with a dataset directory): five PLY frames (ascii and binary, shuffled
    vertex order, duplicated points - what read_ply_o3d + the voxel de-duplication of custom_dataset.py:259-270 accept), GOPs of
    2 (the last one a single frame), every frame decoded from the written files and compared with the de-duplicated input."""
    from linr_pcgc_amd import ply, run, synthetic
    rng = np.random.default_rng(5)
    files, clouds = [], []
    for t in range(5):
        xyz = synthetic.sphere_shell(7, 38 + t, centre=(64 + t, 60, 66))
        clouds.append(xyz)
        shuffled = np.concatenate([xyz, xyz[:100]], axis=0)[rng.permutation(len(xyz) + 100)]
        path = str(tmp_path / ('frame_%04d.ply' % t))
        ply.write_ply_xyz(path, shuffled, binary=bool(t % 2))
        files.append(path)
    out = str(tmp_path / 'seq_ply')
    # the reference's spellings of the flags (main.py:480-534)
    args = run.parse(['--ori_dir', str(tmp_path), '--ori_dtype', 'ply', '--frame_num', '9', '--gop_size', '2', '--first_epoch', '2',
                      '--others_epoch', '1', '--result_dir', out, '--min_point_num', '64', '--model_bitdepth', '8', '--decode'])
    assert run.resolve_files(args) == files and args.frames == 5
    summary, results = run.run_sequence_job(args, 0, 1, None, files=files)
    assert summary['gops'] == 3 and summary['lossless'] is True and sorted(results) == [0, 1, 2]
    assert [results[g]['frames'] for g in range(3)] == [2, 2, 1]
    assert sum(r['points'] for r in results.values()) == sum(len(c) for c in clouds)          # duplicates dropped, nothing else
    # and once more from the files alone, like decoder.py: GOP 2 (one frame)
    from linr_pcgc_amd import codec, overfit
    enc = codec.read_gop(os.path.join(out, 'result_enc', 'gop_4_4'))
    dec = codec.decode_gop(overfit.gen_model(len(enc['frames'][0]), 'cuda'), enc, 'cuda', workers=1)
    assert np.array_equal(dec[0].cpu().numpy(), clouds[4])


def test_config3_andrew10_two_gop_sequence(pkg, tmp_path):
    """BASELINE config[3] in miniature on one GPU: the MVUB andrew10 stand-in (10-bit 2-voxel-thick shell, 1.3 M points,
    K_eff 16-18: the densest kernel map of the configs), 64 frames in GOPs of 32, one epoch each: GOP 1 warm-starts from
    GOP 0's checkpoint, both are encoded to files, 3 frames per GOP decoded from the files and compared bit for bit."""
    from linr_pcgc_amd import run
    out = str(tmp_path / 'seq3')
    args = run.parse(['--config', 'andrew10', '--frames', '64', '--gop', '32', '--first-epoch', '1', '--others-epoch', '1',
                      '--out', out, '--decode'])
    summary, results = run.run_sequence_job(args, 0, 1, None, decode_frames=3)
    assert summary['gops'] == 2 and summary['lossless'] is True and sorted(results) == [0, 1]
    assert results[0]['points'] > 32 * 1250000 and all(r['lossless'] for r in results.values())
    assert results[1]['loss'][-1] < 0.8 * results[0]['loss'][-1]          # warm start
