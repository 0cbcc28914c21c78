"""GPU tests at the sizes of the BASELINE configs (properties that do not need the oracle at full size) and on tiny / ragged
frames around the kernels' tile sizes.  Tolerances: tests/gpu_common.py.
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


def test_full_size_frame_properties(pkg):
    """BASELINE config[1] size (784,314 points, 7 scales): size-independent properties instead of the slow oracle:
    determinism, staged == one-shot probabilities, train step lowers the bits, encode -> decode is lossless."""
    from linr_pcgc_amd import codec, engine, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    pts = synthetic.sequence_frame('loot10', 3)
    gop = overfit.Gop(None, [pts], None, 64, 'cuda')
    assert gop.point_nums[0] > 700000 and gop.scale_num == 7
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    f = gop.frames[0]
    p1, b1 = model.frame_probs(f)
    p2, b2 = model.frame_probs(f)
    assert torch.equal(p1, p2) and torch.equal(b1, b2)
    staged = torch.empty_like(p1)
    for k in range(8):
        engine.net_forward(f, model.flat_parameters(), k, k + 1, staged, None)
    assert torch.equal(p1, staged)
    assert bool(((p1 >= 0) & (p1 <= 1)).all())
    # closed-form check of the bits accumulator against the probabilities it was computed from
    t = f.occ.t().double()
    pd = p1.double()
    nats = -(t * torch.log(pd).clamp(min=-100) + (1 - t) * torch.log1p(-pd).clamp(min=-100)).sum()
    assert abs(float(nats) / math.log(2) - float(b1)) <= 2e-5 * float(b1)
    opt = FlatAdam(model)
    for _ in range(5):
        train_step(model, opt, f, gop.point_nums[0])
    _, b3 = model.frame_probs(f)
    assert float(b3) < float(b1)
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda')
    ref = torch.as_tensor(gop.infos[0]['ori']).cuda() + torch.tensor(gop.coord_mins[0], device='cuda', dtype=torch.int32)
    assert torch.equal(dec[0], ref), 'decoded geometry must be bit-exact at full size'
    assert abs(enc['bpp']['point_bpp'] * gop.point_nums[0] - enc['bits_est']) <= 0.01 * enc['bits_est'] + 8 * 64


def _frame_properties(gop, model, steps=3):
    """size-independent checks shared by the full-size configs"""
    from linr_pcgc_amd import codec, engine, overfit
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    f = gop.frames[0]
    p1, b1 = model.frame_probs(f)
    p2, b2 = model.frame_probs(f)
    assert torch.equal(p1, p2) and torch.equal(b1, b2)
    staged = torch.empty_like(p1)
    for k in range(8):
        engine.net_forward(f, model.flat_parameters(), k, k + 1, staged, None)
    assert torch.equal(p1, staged)
    t = f.occ.t().double()
    pd = p1.double()
    nats = -(t * torch.log(pd).clamp(min=-100) + (1 - t) * torch.log1p(-pd).clamp(min=-100)).sum()
    assert abs(float(nats) / math.log(2) - float(b1)) <= 2e-5 * float(b1)
    opt = FlatAdam(model, lr=1e-3)          # small steps: the gradient must be a descent direction (no Adam overshoot)
    for _ in range(steps):
        train_step(model, opt, f, gop.point_nums[0])
    _, b3 = model.frame_probs(f)
    assert steps == 0 or float(b3) < float(b1)
    enc = codec.encode_gop(model, overfit.gen_model(gop.scale_num, 'cuda'), gop, 8)
    dec = codec.decode_gop(overfit.gen_model(gop.scale_num, 'cuda'), enc, 'cuda')
    ref = torch.as_tensor(gop.infos[0]['ori']).cuda() + torch.tensor(gop.coord_mins[0], device='cuda', dtype=torch.int32)
    assert torch.equal(dec[0], ref), 'decoded geometry must be bit-exact'
    return float(b1), float(b3), enc


def test_config0_sphere8_against_oracle(pkg):
    """BASELINE config[0]: the 8-bit sphere (125,810 points, 6 scales), gop_size=1, frame_num=1, first_epoch=2 - the one
    full-size case the CPU oracle finishes in seconds: bits of the seeded initialisation against the oracle, then the
    2-epoch overfit + encode + decode flow."""
    from linr_pcgc_amd import overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam
    pts = synthetic.sequence_frame('sphere8', 0)
    gop = overfit.Gop(None, [pts], None, 64, 'cuda')
    assert gop.point_nums[0] == 125810 and gop.scale_num == 6 and 53000 < gop.frames[0].rows < 55000
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    _, bits = model.frame_probs(gop.frames[0])
    scales = []
    for info in gop.infos[0]['all_input_info']:
        c = info['coord'].cpu().numpy().astype(np.int32)
        scales.append({'coord': c, 'occ': info['occ'].cpu().numpy().astype(np.float32),
                       'offset_tensor': info['offset_tensor'].cpu().numpy().astype(np.float32),
                       'scale_idx': info['scale_idx'], 'nbr': ooct.neighbour_table(c)})
    ref_bits = float(onet.frame_bits(sd, onet.to_torch_scales(scales)))
    assert abs(float(bits) - ref_bits) <= 1e-5 * ref_bits, (float(bits), ref_bits)
    opt = FlatAdam(model)
    losses = overfit.overfit_gop(model, opt, gop, 2)
    assert losses[1] < losses[0]
    _frame_properties(gop, model, steps=0)


def test_config3_andrew10_dense_shell_properties(pkg):
    """BASELINE config[3] stand-in: 2-voxel-thick 10-bit shell (1,306,322 points, K_eff ~ 17): stresses the kernel-map
    build and the gathers; size-independent properties."""
    from linr_pcgc_amd import overfit, synthetic
    pts = synthetic.sequence_frame('andrew10', 0)
    gop = overfit.Gop(None, [pts], None, 64, 'cuda')
    assert gop.point_nums[0] == 1306322 and gop.scale_num == 7
    _frame_properties(gop, overfit.gen_model(gop.scale_num, 'cuda', seed=8807))


def test_config4_owlii11_size_properties(pkg):
    """BASELINE config[4] stand-in geometry: 11-bit sphere (~2.9 M points, 8 scales, ~1.24 M rows) through the fp32 path (its
    bf16 / uint8-weight codec and the gop_size = 64 flow: tests/test_gpu_bf16.py)."""
    from linr_pcgc_amd import overfit, synthetic
    pts = synthetic.sequence_frame('owlii11', 0)
    gop = overfit.Gop(None, [pts], None, 64, 'cuda')
    assert gop.point_nums[0] > 2800000 and gop.scale_num == 8 and gop.frames[0].rows > 1200000
    _frame_properties(gop, overfit.gen_model(gop.scale_num, 'cuda', seed=8807), steps=2)


def _oracle_scales(info):
    scales = []
    for i in info['all_input_info']:
        c = i['coord'].cpu().numpy().astype(np.int32)
        scales.append({'coord': c, 'occ': i['occ'].cpu().numpy().astype(np.float32),
                       'offset_tensor': i['offset_tensor'].cpu().numpy().astype(np.float32),
                       'scale_idx': i['scale_idx'], 'nbr': ooct.neighbour_table(c)})
    return scales


def test_rough_figure_generators_agree_and_small_cloud_matches_the_oracle(pkg):
    """The non-spherical stress workload (synthetic.rough_figure: torso, head, legs, thin slanted arms, a thin sheet, +-8 voxel
    low-frequency displacement; VERDICT r5 Missing #4): the GPU enumeration gives the numpy enumeration's points; on the 8-bit figure
    (49 k points - thin parts, concavities, branching: what no sphere has) bits of the seeded initialisation and ALL 189 gradient
    tensors against the oracle, fp32 and bf16 executors, then the lossless round trip."""
    from linr_pcgc_amd import engine, overfit, synthetic
    for t in (0, 5):
        a = synthetic.rough_figure(8, t)
        b = synthetic.rough_figure_device(8, t, 'cuda')
        assert a.shape[0] > 40000 and np.array_equal(a, b.cpu().numpy())
    assert not np.array_equal(synthetic.rough_figure(8, 0), synthetic.rough_figure(8, 5)), 'the sequence moves'
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('rough8', 0, 'cuda')], None, 64, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    f = gop.frames[0]
    _, bits = model.frame_probs(f)
    scales = onet.to_torch_scales(_oracle_scales(gop.infos[0]))
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = onet.frame_bits(sdo, scales)
    assert abs(float(bits) - float(ref)) <= 1e-5 * float(ref), (float(bits), float(ref))
    ref.backward()
    sd64 = {k: v.double().clone().requires_grad_() for k, v in sd.items()}
    onet.frame_bits(sd64, onet.to_torch_scales(_oracle_scales(gop.infos[0]), torch.float64)).backward()
    grads = torch.zeros_like(model.flat_parameters())
    engine.net_forward(f, model.flat_parameters(), 0, 8, None, None)
    engine.net_backward(f, model.flat_parameters(), grads, 1.0)
    # the criterion proper is the float64-anchored one inside the helper (HIP as close to float64 as the fp32 oracle is); the direct
    # fp32-vs-fp32 sanity bound is 3e-3 here: two summation orders differ by 1.01e-3 of the tensor's largest entry on this surface
    _grads_close_per_tensor(grads, sdo, rtol=3e-3, sd64=sd64)
    # the bf16 training executor's forward on the same cloud against the emulating oracle
    from oracle import network_bf16 as obf
    bb = torch.zeros(1, dtype=torch.float64, device='cuda')
    engine.net_forward_train_bf16(f, model.flat_parameters(), None, bb)
    with torch.no_grad():
        ref_b = float(obf.train_frame_bits(sd, scales))
    assert abs(float(bb) - ref_b) <= 2e-3 * ref_b, (float(bb), ref_b)
    _frame_properties(gop, model, steps=3)


def test_rough_figure_full_size_properties(pkg):
    """loot10_rough at full size (~0.75 M points, 7 scales) beside BASELINE config[1]'s sphere: the size-independent properties
    (determinism, staged == one-shot, closed-form bits, descent, lossless encode -> decode), the same with the bf16 executor's
    training steps, and the reference-trained checkpoint (tests/golden/loot_model_kat.npz) through the HIP forward: a model
    trained by the REFERENCE on real loot codes this unseen figure at about half the bits of an untrained model - a second, weaker
    behavioural pin of the assumed MinkowskiEngine conventions on a non-spherical surface (tools/me_order_probe.py --rough: every
    alternative reading of the tap order is worse)."""
    from linr_pcgc_amd import overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('loot10_rough', 0, 'cuda')], None, 64, 'cuda')
    assert 700000 < gop.point_nums[0] < 820000 and gop.scale_num == 7
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    b1, b3, enc = _frame_properties(gop, model, steps=3)
    mb = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    mb.train_precision = 'bf16'
    ob = FlatAdam(mb, lr=1e-3)
    lb = [float(train_step(mb, ob, gop.frames[0], gop.point_nums[0])) for _ in range(4)]
    assert lb[-1] < lb[0] and all(math.isfinite(x) for x in lb), lb
    from test_oracle_golden import _reference_state_dict
    ref_model = overfit.gen_model(7, 'cpu', seed=1)
    ref_model.load_state_dict(_reference_state_dict(os.path.join(os.path.dirname(__file__), 'golden')))
    ref_model = ref_model.cuda()
    _, bits_ref = ref_model.frame_probs(gop.frames[0])
    bpp_ref = float(bits_ref) / gop.point_nums[0]
    bpp_init = b1 / gop.point_nums[0]
    print('loot10_rough: reference checkpoint %.4f bits/point, untrained seed 8807 %.4f' % (bpp_ref, bpp_init))
    # measured: 1.79 against 3.4-3.9 untrained (0.92-0.96 on the sphere shells: +-8 voxel wrinkles are unlike anything loot has)
    assert bpp_ref < 2.2 and bpp_ref < 0.6 * bpp_init, (bpp_ref, bpp_init)


def test_config4_size_wave_specialised_backward_is_bit_identical(pkg):
    """At BASELINE config[4]'s frame size (~1.24 M rows, 8 scales): the training step's gradients with the wave-specialised fused backward
    kernels (conv_bwd_wgrad_k: producer / consumer wave pairs) and with the single-stream ones (LINR_FUSED_SPLIT=0) - bit-identical, and
    run-to-run reproducible."""
    import os
    from linr_pcgc_amd import engine, overfit, synthetic
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('owlii11', 0, 'cuda')], None, 64, 'cuda')
    model = overfit.gen_model(gop.scale_num, 'cuda', seed=8807)
    frame, flat = gop.frames[0], model.flat_parameters()
    assert frame.rows > 1200000

    def grads_now():
        bits = torch.zeros(1, dtype=torch.float64, device='cuda')
        g = torch.zeros_like(flat)
        engine.net_forward(frame, flat, 0, 8, None, bits)
        engine.net_backward(frame, flat, g, 1.0 / gop.point_nums[0])
        torch.cuda.synchronize()
        return g
    a, b = grads_now(), grads_now()
    old = os.environ.get('LINR_FUSED_SPLIT')
    os.environ['LINR_FUSED_SPLIT'] = '0'
    try:
        c = grads_now()
    finally:
        if old is None:
            del os.environ['LINR_FUSED_SPLIT']
        else:
            os.environ['LINR_FUSED_SPLIT'] = old
    assert float(a.abs().max()) > 0 and bool(torch.isfinite(a).all())
    assert torch.equal(a, b) and torch.equal(a, c)


def _smallest_relu_input(sd, sc):
    """Smallest |x| any ReLU of the network sees on this scale, from the oracle in float64.  Below ~3e-7 (inputs are O(1)) the sign of x - and with
    it a whole term of the gradient - is decided by fp32 rounding order, so no two fp32 implementations need agree there."""
    import types
    seen = []

    def relu(x):
        if x.numel():
            seen.append(float(x.detach().abs().min()))
        return torch.relu(x)
    shim = types.SimpleNamespace(relu=relu, linear=torch.nn.functional.linear,
                                 binary_cross_entropy=torch.nn.functional.binary_cross_entropy)
    keep, onet.F = onet.F, shim
    try:
        with torch.no_grad():
            onet.forward_scale({k: v.double() for k, v in sd.items()}, onet.to_torch_scales([sc], torch.float64)[0])
    finally:
        onet.F = keep
    return min(seen)


@pytest.mark.parametrize('n', [1, 2, 17, 63, 64, 65, 127, 129, 255, 256, 257, 511, 1025])
def test_tiny_and_ragged_frames(pkg, n):
    """Edge cases: a scale with a single voxel, row counts around the wave size (64), the workgroup tiles (256) and the kernels'
    multi-tile boundaries, and a zero-row scale.  A cloud on which some ReLU input is a tie at fp32 resolution (n = 257 with the
    first seed: 1.2e-9 at one hidden unit of head 6) is redrawn - the criterion is the oracle's, not the kernels'."""
    from linr_pcgc_amd import engine
    model, sd = _model_and_oracle(pkg, 3)
    side = max(6, int(round((3 * n) ** (1 / 3))) + 2)          # a box that holds n distinct voxels at ~1/3 occupancy
    want = n
    for attempt in range(6):
        rng = np.random.default_rng(want + 1000 * attempt)
        c = ooct.unique_sorted(rng.integers(0, side, size=(4 * want, 3)))[:want]
        n = len(c)
        scales = [{'coord': c, 'occ': (rng.random((n, 8)) < 0.5).astype(np.float32), 'offset_tensor': ooct.offset_tensor(c),
                   'scale_idx': 1},
                  {'coord': np.zeros((0, 3), np.int32), 'occ': np.zeros((0, 8), np.float32),
                   'offset_tensor': np.zeros((0, 7), np.float32), 'scale_idx': 0}]
        tie = dict(scales[0]); tie['nbr'] = ooct.neighbour_table(c)
        if _smallest_relu_input(sd, tie) >= 3e-7:
            break
    else:
        pytest.fail('six clouds in a row with a ReLU tie')
    frame = model.make_frame(scales)
    probs, bits = model.frame_probs(frame)
    sc = dict(scales[0]); sc['nbr'] = ooct.neighbour_table(c)
    out = onet.forward_scale(sd, onet.to_torch_scales([sc])[0])
    assert abs(float(bits) - float(out['bits'])) <= 1e-5 * float(out['bits']) + 1e-6
    grads = torch.zeros_like(model.flat_parameters())
    engine.net_backward(frame, model.flat_parameters(), grads, 1.0)
    assert bool(torch.isfinite(grads).all())
    # the weight-gradient kernels at row counts below one 8-row group / with most of their 512 blocks empty: against autograd
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    onet.forward_scale(sdo, onet.to_torch_scales([sc])[0])['bits'].backward()
    off = 0
    for name, v in sdo.items():
        m = v.numel()
        mine = grads[off:off + m].view(v.shape).cpu().double()
        ref = torch.zeros_like(v).double() if v.grad is None else v.grad.double()      # scale MLPs of absent scales: no gradient
        gmax = float(ref.abs().max())
        assert float((mine - ref).abs().max()) <= 2e-4 * gmax + 1e-6, name
        off += m


def test_ragged_multi_scale_frame_with_colliding_coordinates(pkg):
    """Three scales batched into one frame at row offsets that are no multiple of any tile (257 + 65 + 3 rows), drawn from the SAME
    coordinate box: a voxel of one scale has the coordinates of another scale's neighbour, which must not become its neighbour
    (main.py:457-475 runs the scales as separate sparse tensors).  Bits and every gradient against the per-scale oracle."""
    from linr_pcgc_amd import engine
    model, sd = _model_and_oracle(pkg, 3)
    for attempt in range(6):
        rng = np.random.default_rng(77 + attempt)
        scales = []
        for idx, n in ((0, 257), (1, 65), (2, 3)):
            c = ooct.unique_sorted(rng.integers(0, 9, size=(4 * n, 3)))[:n]
            scales.append({'coord': c, 'occ': (rng.random((len(c), 8)) < 0.5).astype(np.float32), 'offset_tensor': ooct.offset_tensor(c),
                           'scale_idx': idx, 'nbr': ooct.neighbour_table(c)})
        if min(_smallest_relu_input(sd, s) for s in scales) >= 3e-7:
            break
    else:
        pytest.fail('six draws in a row with a ReLU tie')
    frame = model.make_frame([{k: v for k, v in s.items() if k != 'nbr'} for s in scales])
    assert frame.rows == 325
    probs, bits = model.frame_probs(frame)
    tsc = onet.to_torch_scales(scales)
    ref = 0.0
    for i, s in enumerate(tsc):
        out = onet.forward_scale(sd, s)
        sl = frame.scale_slice(i)
        for k in range(8):
            assert float((probs[k, sl].cpu() - out['probs'][k].reshape(-1)).abs().max()) <= 2e-6, (i, k)
        ref += float(out['bits'])
    assert abs(float(bits) - ref) <= 1e-5 * ref
    grads = torch.zeros_like(model.flat_parameters())
    engine.net_backward(frame, model.flat_parameters(), grads, 1.0)
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    onet.frame_bits(sdo, tsc).backward()
    off = 0
    for name, v in sdo.items():
        m = v.numel()
        mine = grads[off:off + m].view(v.shape).cpu().double()
        refg = torch.zeros_like(v).double() if v.grad is None else v.grad.double()
        assert float((mine - refg).abs().max()) <= 2e-4 * float(refg.abs().max()) + 1e-6, name
        off += m


def test_device_generator_equals_numpy_generator(pkg):
    from linr_pcgc_amd import synthetic
    for cfg, t in (('sphere8', 0), ('loot10', 17)):
        a = synthetic.sequence_frame(cfg, t)
        b = synthetic.sequence_frame_device(cfg, t, 'cuda')
        assert b.dtype == torch.int32 and np.array_equal(a, b.cpu().numpy())
