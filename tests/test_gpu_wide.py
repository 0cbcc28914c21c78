"""hidden_channel_conv = 16 (main.py:520; models/upsample.py:38-76, models/resnet.py:12-51): the channel-blocked executor
(linr_pcgc_amd/wide_net.py) against the oracle - forward bits and probabilities, every parameter gradient against autograd,
training steps against torch.optim.Adam on the oracle, and a lossless encode / decode round trip through the codec."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import network as onet          # noqa: E402
from oracle import octree as ooct           # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pkg():
    import linr_pcgc_amd
    from linr_pcgc_amd import _lib
    _lib.lib()
    return linr_pcgc_amd


@pytest.fixture(scope='module')
def shell(golden_dir):
    g = np.load(os.path.join(golden_dir, 'octree_shell128.npz'))
    scales = []
    for s in range(int(g['scale_num'])):
        c = g['s%d_coord' % s]
        scales.append({'coord': c, 'occ': g['s%d_occ' % s], 'offset_tensor': g['s%d_offset' % s], 'scale_idx': s,
                       'nbr': ooct.neighbour_table(c)})
    return {'scales': scales, 'point_num': int(len(g['ori']))}


def _model(hidden, scale_num=5, block_layers=1, seed=8807):
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    torch.manual_seed(seed)
    model = LINR_PCGC_Model({'scale_num': scale_num, 'in_channel': 7, 'hidden_channel_conv': hidden, 'block_layers': block_layers,
                             'outstage': 8, 'instage': 1})
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.cuda(), sd


def test_wide_state_dict_has_the_reference_shapes(pkg):
    """Names and shapes follow models/upsample.py:38-76 / models/resnet.py:12-51 with channels = 16."""
    model, sd = _model(16)
    assert sd['upsampler.block_in.0.kernel'].shape == (27, 8, 16)
    assert sd['upsampler.block_in.2.layers.0.conv0_0.kernel'].shape == (27, 16, 8)
    assert sd['upsampler.block_in.2.layers.0.conv1_0.kernel'].shape == (16, 8)
    assert sd['upsampler.block_in.2.layers.0.conv1_2.kernel'].shape == (8, 8)
    assert sd['upsampler.block_in.3.kernel'].shape == (27, 16, 16)
    assert sd['upsampler.inner_mlps.3.0.0.weight'].shape == (24, 16)
    assert sd['upsampler.prune_blocks.5.0.conv.kernel'].shape == (27, 16, 16)
    assert sd['upsampler.outter_blocks.2.0.kernel'].shape == (27, 3, 16)
    assert len(sd) == 181          # scale_num 5: 1 + 4 x 5 + 14 + 8 x 4 + 8 x 2 + 7 x 14 (189 at scale_num 7), as at width 8
    with pytest.raises(ValueError):
        _model(12)


@pytest.mark.parametrize('hidden,block_layers', [(16, 1), (16, 2), (32, 1), (32, 3)])
def test_wide_forward_and_backward_match_the_oracle(pkg, shell, hidden, block_layers):
    model, sd = _model(hidden, block_layers=block_layers)
    frame = model.make_frame(shell['scales'])
    probs, bits = model.frame_probs(frame)
    tsc = onet.to_torch_scales(shell['scales'])
    ref_bits = 0.0
    for i, s in enumerate(tsc):
        out = onet.forward_scale(sd, s)
        sl = frame.scale_slice(i)
        for k in range(8):
            err = (probs[k, sl].double().cpu() - out['probs'][k].reshape(-1).double()).abs().max()
            assert float(err) <= 1e-4, (i, k, float(err))
        ref_bits += float(out['bits'])
    assert abs(float(bits) - ref_bits) <= 1e-5 * ref_bits, (float(bits), ref_bits)
    probs2, bits2 = model.frame_probs(frame)
    assert torch.equal(probs, probs2) and torch.equal(bits, bits2), 'forward must be bit-reproducible'
    # gradients through the executor's own tape against autograd of the oracle, every tensor against its own largest entry
    gscale = 1.0 / shell['point_num']
    model._ensure_grad_views()
    model._flat_grad.zero_()
    b = torch.zeros(1, dtype=torch.float64, device='cuda')
    with torch.no_grad():
        tape = model._wide.forward(frame, 0, 8, None, b, keep=True)
        model._wide.backward(frame, tape, gscale)
    grads = model._flat_grad.clone()
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    (onet.frame_bits(sdo, tsc) * gscale).backward()
    off = 0
    for name, v in sdo.items():
        n = v.numel()
        mine = grads[off:off + n].view(v.shape).double().cpu()
        ref = torch.zeros_like(v).double() if v.grad is None else v.grad.double()
        gmax = float(ref.abs().max())
        assert float((mine - ref).abs().max()) <= 3e-3 * gmax + 1e-9, name
        off += n
    with torch.no_grad():
        model._flat_grad.zero_()
        tape = model._wide.forward(frame, 0, 8, None, b, keep=True)
        model._wide.backward(frame, tape, gscale)
    assert torch.equal(grads, model._flat_grad), 'backward must be bit-reproducible'


def test_wide_model_surface_autograd(pkg, shell):
    """loss = model(putin_args); loss.backward() like main.py:457-475, one scale."""
    model, sd = _model(16)
    s = shell['scales'][1]
    dev = 'cuda'
    inargs = {'coord': torch.as_tensor(s['coord']).to(dev), 'offset_tensor': torch.as_tensor(s['offset_tensor']).to(dev),
              'occ_lst': [torch.as_tensor(s['occ'][:, k:k + 1].copy()).to(dev) for k in range(8)], 'scale_idx': 1}
    loss = model(inargs) / 1000.0
    loss.backward()
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = onet.forward_scale(sdo, onet.to_torch_scales([s])[0])['bits'] / 1000.0
    assert abs(float(loss) - float(ref)) <= 1e-5 * float(ref)
    ref.backward()
    for (name, p) in model.named_parameters():
        g = sdo[name].grad
        g = torch.zeros_like(sdo[name]) if g is None else g
        gmax = float(g.abs().max())
        assert float((p.grad.cpu().double() - g.double()).abs().max()) <= 3e-3 * gmax + 1e-9, name


def test_wide_train_steps_track_torch_adam_and_decode_losslessly(pkg, shell):
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    model, sd = _model(16)
    frame = model.make_frame(shell['scales'])
    opt = FlatAdam(model)
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    opt_o = torch.optim.Adam(list(sdo.values()), lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    tsc = onet.to_torch_scales(shell['scales'])
    pn = shell['point_num']
    first = None
    for it in range(4):
        bits = train_step(model, opt, frame, pn)
        opt_o.zero_grad()
        b_o = onet.frame_bits(sdo, tsc)
        (b_o / pn).backward()
        opt_o.step()
        assert abs(float(bits) - float(b_o)) <= 2e-4 * float(b_o), (it, float(bits), float(b_o))
        first = float(bits) if first is None else first
    assert first > 0          # (four Adam steps at lr 0.01 need not lower the loss yet; the oracle's trajectory is the criterion)
    # encoder probabilities == the staged decoder's, decoded occupancy == the input
    probs, _ = model.frame_probs(frame)
    from linr_pcgc_amd.model_core import encode_streams
    p_host = probs.cpu().numpy()
    occ = np.concatenate([s['occ'] for s in shell['scales']], axis=0)
    streams = []
    for i in range(frame.n_scales):
        sl = frame.scale_slice(i)
        streams.append(encode_streams([p_host[k, sl] for k in range(8)], [occ[sl, k].astype(np.uint8) for k in range(8)]))
    dec_frame = model.make_frame([{k: v for k, v in s.items() if k != 'occ'} for s in shell['scales']])
    out = model.decode_frame(dec_frame, streams)
    got = torch.cat(out, dim=1).cpu().numpy()
    assert np.array_equal(got, occ)


def test_wide_gop_encode_decode_roundtrip(pkg):
    """A 2-frame GOP of the 8-bit sphere through overfit -> encode_gop -> decode_gop with hidden_channel_conv = 16: lossless."""
    from linr_pcgc_amd import codec, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam, LINR_PCGC_Model
    clouds = [synthetic.sequence_frame_device('sphere8', t, 'cuda') for t in range(2)]
    gop = overfit.Gop(None, clouds, None, 64, 'cuda')

    def gen(seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        return LINR_PCGC_Model({'scale_num': gop.scale_num, 'in_channel': 7, 'hidden_channel_conv': 16, 'block_layers': 1,
                                'outstage': 8, 'instage': 1}).cuda()
    model = gen(8807)
    losses = overfit.overfit_gop(model, FlatAdam(model), gop, 3)
    assert losses[-1] < losses[0]
    enc = codec.encode_gop(model, gen(), gop, 8)
    dec = codec.decode_gop(gen(), enc, 'cuda', workers=1)
    for i in range(2):
        ref = torch.as_tensor(gop.infos[i]['ori']).cuda() + torch.tensor(gop.coord_mins[i], device='cuda', dtype=torch.int32)
        assert torch.equal(dec[i], ref)


@pytest.mark.parametrize('cin,cout', [(16, 16), (8, 16), (16, 8), (32, 32), (3, 16), (16, 32)])
def test_wide_conv_entries_match_the_oracle_conv(pkg, shell, cin, cout):
    """linr_spconv_wide (forward with bias / residual / ReLU, backward-data with the ReLU mask) and linr_spconv_wgrad_wide against
    oracle.network.conv3 (MinkowskiConvolution kernel_size 3) and its autograd on the finest scale of the
    shell.  Tolerance: fp32 sums of up to 27 x 32 products in a different order (1e-5 relative to the largest entry)."""
    from linr_pcgc_amd import ops
    sc = shell['scales'][0]
    n = len(sc['coord'])
    nbr_o = torch.from_numpy(sc['nbr']).long().cuda()                          # oracle layout [n, 27]
    ld = (n + 63) // 64 * 64
    nbr = torch.full((27, ld), -1, dtype=torch.int32, device='cuda')           # library layout [27, ld]
    nbr[:, :n] = ops.kmap_build(torch.from_numpy(sc['coord']).cuda())
    lo, mask = ops.kmap_compress(nbr, n)
    tile8t = ops.kmap_tile8t(nbr, n)
    torch.manual_seed(cin * 100 + cout)
    W = (torch.randn(27, cin, cout, device='cuda') * 0.1).requires_grad_()
    b = torch.randn(1, cout, device='cuda', requires_grad=True)
    nbi, nbo = (cin + 7) // 8, cout // 8

    def blocks(t, nb):
        """[n, 8 nb] (zero-padded channels) -> blocks: views buf[i, 1:] of [n + 1, 8] buffers whose row 0 is zero."""
        buf = torch.zeros((nb, n + 1, 8), device='cuda')
        for i in range(nb):
            w = min(8, t.shape[1] - 8 * i)
            buf[i, 1:, :w] = t[:, 8 * i:8 * i + w]
        return buf, [buf[i, 1:] for i in range(nb)]

    x = torch.randn(n, cin, device='cuda', requires_grad=True)
    res = torch.randn(n, cout, device='cuda')
    xbuf, xs = blocks(x.detach(), nbi)
    if cin % 8:
        xbuf[-1, 1:, cin % 8:] = 7.0                                           # channels past cin must be ignored
    _, rs = blocks(res, nbo)
    outs = ops.spconv_wide(xs, lo, mask, n, W.detach(), b.detach().reshape(-1), res=rs, relu=True)
    pre = onet.conv3(x, nbr_o, W, b) + res
    ref = torch.relu(pre)
    got = torch.cat(outs, dim=1)
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())

    g = torch.randn(n, cout, device='cuda')
    _, gs = blocks(g, nbo)
    y = onet.conv3(x, nbr_o, W, b)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y, [x, W, b], g)
    gw, gb = ops.spconv_wgrad_wide(xs, gs, nbr, tile8t, n, cin, cout)
    assert float((gw - gw_ref).abs().max()) <= 1e-5 * float(gw_ref.abs().max())
    assert float((gb - gb_ref.reshape(-1)).abs().max()) <= 1e-5 * float(gb_ref.abs().max())
    if cin % 8 == 0:
        act = torch.randn(n, cin, device='cuda')
        _, acts = blocks(act, nbi)
        gi = ops.spconv_wide(gs, lo, mask, n, W.detach(), None, bwd=True, act=acts)
        want = gx_ref * (act > 0)
        assert float((torch.cat(gi, dim=1) - want).abs().max()) <= 1e-5 * float(want.abs().max())


def _to_blocks(t, padded=True):
    """[n, 8 nb] -> nb blocks [n, 8] (views buf[i, 1:] of [n + 1, 8] buffers whose row 0 is zero when padded)."""
    n, nb = t.shape[0], t.shape[1] // 8
    buf = torch.zeros((nb, n + 1, 8), device=t.device)
    for i in range(nb):
        buf[i, 1:] = t[:, 8 * i:8 * i + 8]
    return [buf[i, 1:] for i in range(nb)]


@pytest.mark.parametrize('cin,cout,layout', [(16, 8, 'me'), (32, 16, 'me'), (8, 8, 'me'), (16, 16, 'me'), (16, 24, 'torch'), (32, 24, 'torch')])
def test_wide_pointwise_entries_match_torch(pkg, cin, cout, layout):
    """linr_linear_wide (forward with bias / residual / ReLU; backward-data with accumulation and the ReLU mask) and
    linr_linear_wgrad_wide on blocked activations against torch matmuls and autograd (MinkowskiConvolution kernel_size 1:
    models/resnet.py:25-46; nn.Linear of the head: models/upsample.py:73-76).  n = 1000: not a multiple of any tile.  Tolerance: fp32
    sums of <= 32 products in another order (1e-5 of the largest entry); the weight gradients sum 1000 rows (1e-4)."""
    from linr_pcgc_amd import ops
    n = 1000
    torch.manual_seed(7 * cin + cout)
    x = torch.randn(n, cin, device='cuda', requires_grad=True)
    W = (torch.randn((cin, cout) if layout == 'me' else (cout, cin), device='cuda') * 0.3).requires_grad_()
    b = torch.randn(cout, device='cuda', requires_grad=True)
    ws = (cout, 1) if layout == 'me' else (1, cin)
    blocked_out = cout % 8 == 0 and layout == 'me'
    xs = _to_blocks(x.detach())
    res = torch.randn(n, cout, device='cuda')
    y_ref = x @ (W if layout == 'me' else W.t()) + b
    outs = _to_blocks(torch.zeros(n, cout, device='cuda')) if blocked_out else [torch.empty(n, cout, device='cuda')]
    ops.linear_wide(xs, cin, W.detach(), ws[0], ws[1], b.detach(), cout, outs, out_blocked=blocked_out,
                    res=_to_blocks(res) if blocked_out else [res], relu=True)
    want = torch.relu(y_ref + res)
    got = torch.cat(outs, dim=1)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # backward-data (+ old, masked) and the weight gradients
    g = torch.randn(n, cout, device='cuda')
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, [x, W, b], g)
    gs = _to_blocks(g) if blocked_out else [g]
    old, act = torch.randn(n, cin, device='cuda'), torch.randn(n, cin, device='cuda')
    gins = _to_blocks(old)
    ops.linear_wide(gs, cout, W.detach(), ws[1], ws[0], None, cin, gins, in_blocked=blocked_out, act=_to_blocks(act), accumulate=True)
    want = (gx_ref + old) * (act > 0)
    assert float((torch.cat(gins, dim=1) - want).abs().max()) <= 1e-5 * float(want.abs().max())
    gw, gb = torch.full_like(W, float('nan')), torch.full_like(b, float('nan'))
    ops.linear_wgrad_wide(xs, cin, gs, cout, gw, ws[0], ws[1], gb, g_blocked=blocked_out)
    assert float((gw - gw_ref).abs().max()) <= 1e-4 * float(gw_ref.abs().max())
    assert float((gb - gb_ref).abs().max()) <= 1e-4 * float(gb_ref.abs().max())
    ops.linear_wgrad_wide(xs, cin, gs, cout, gw, ws[0], ws[1], gb, g_blocked=blocked_out, accumulate=True)       # LINR_ACCUM
    assert float((gw - 2 * gw_ref).abs().max()) <= 2e-4 * float(gw_ref.abs().max())


@pytest.mark.parametrize('C,stages,n', [(16, 8, 1000), (32, 3, 777), (16, 1, 64 * 300 + 5)])
def test_wide_head_entries_match_torch(pkg, C, stages, n):
    """linr_head_wide_fwd (p and the stage's bits) and linr_head_wide_bwd (all stages in one grouped launch: gc and the parameter
    gradients [W1 | b1 | w2 | b2] per stage) against torch: Linear(C, 24) -> ReLU -> Linear(24, 1) -> sigmoid -> BCELoss(sum) / ln 2
    (models/upsample.py:137-161, model_core.py:72-81) and its autograd.  Tolerances: p 1e-6 absolute, bits 1e-6 relative, gradients
    1e-4 of the largest entry (sums over n rows in another order)."""
    from linr_pcgc_amd import ops
    torch.manual_seed(C + stages)
    occ = (torch.rand(n, 8, device='cuda') < 0.4).float()
    per = 24 * C + 49
    grads = torch.full((stages * per,), float('nan'), device='cuda')
    cs, ps, gcs, params, refs = [], [], [], [], []
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    bits_ref = 0.0
    gscale = 0.37
    for k in range(stages):
        c = torch.randn(n, C, device='cuda', requires_grad=True)
        w1 = (torch.randn(24, C, device='cuda') * 0.3).requires_grad_()
        b1 = (torch.randn(24, device='cuda') * 0.3).requires_grad_()
        w2 = (torch.randn(1, 24, device='cuda') * 0.3).requires_grad_()
        b2 = (torch.randn(1, device='cuda') * 0.3).requires_grad_()
        z = torch.relu(c @ w1.t() + b1) @ w2.t() + b2
        p_ref = torch.sigmoid(z).reshape(-1)
        nats = torch.nn.functional.binary_cross_entropy(p_ref, occ[:, k], reduction='sum')
        bits_ref += float(nats) / np.log(2.0)
        refs.append(torch.autograd.grad(nats * gscale, [c, w1, b1, w2, b2]))
        blocks = _to_blocks(c.detach())
        p = torch.empty(n, device='cuda')
        ops.head_wide_fwd(blocks, w1.detach(), b1.detach(), w2.detach(), b2.detach(), occ[:, k], p, bits)
        assert float((p - p_ref).abs().max()) <= 1e-6
        cs.append(blocks); ps.append(p); gcs.append(_to_blocks(torch.zeros(n, C, device='cuda')))
        params.append((w1.detach(), b1.detach(), w2.detach()))
    assert abs(float(bits) - bits_ref) <= 1e-6 * bits_ref
    p_only = torch.empty(n, device='cuda')                      # without a target: probabilities only (the decoder's call)
    ops.head_wide_fwd(cs[0], params[0][0], params[0][1], params[0][2], torch.zeros(1, device='cuda'), None, p_only)
    ops.head_wide_bwd(cs, ps, [occ[:, k] for k in range(stages)], [q[0] for q in params], [q[1] for q in params], [q[2] for q in params],
                      gscale, gcs, grads)
    for k in range(stages):
        gc_ref, gw1, gb1, gw2, gb2 = refs[k]
        got = torch.cat(gcs[k], dim=1)
        assert float((got - gc_ref).abs().max()) <= 1e-5 * float(gc_ref.abs().max())
        want = torch.cat([gw1.reshape(-1), gb1, gw2.reshape(-1), gb2])
        assert float((grads[k * per:(k + 1) * per] - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_wide_deferred_reductions_equal_the_immediate_ones(pkg, shell):
    """linr_wide_reduce_many over the partials that linr_spconv_wgrad_wide / linr_linear_wgrad_wide leave behind with gW = NULL gives the
    bits of the entries' own reductions (same slab, same fixed-order sums), for several layers in ONE launch."""
    from linr_pcgc_amd import ops
    sc = shell['scales'][0]
    n = len(sc['coord'])
    ld = (n + 63) // 64 * 64
    nbr = torch.full((27, ld), -1, dtype=torch.int32, device='cuda')
    nbr[:, :n] = ops.kmap_build(torch.from_numpy(sc['coord']).cuda())
    tile8t = ops.kmap_tile8t(nbr, n)
    torch.manual_seed(5)
    deferred, want, got = [], [], []
    for cin, cout in [(16, 16), (3, 16), (32, 8), (16, 32)]:
        xs, gs = _to_blocks(torch.randn(n, (cin + 7) // 8 * 8, device='cuda')), _to_blocks(torch.randn(n, cout, device='cuda'))
        want.append(ops.spconv_wgrad_wide(xs, gs, nbr, tile8t, n, cin, cout))
        gw, gb = torch.full((27, cin, cout), float('nan'), device='cuda'), torch.full((cout,), float('nan'), device='cuda')
        ops.spconv_wgrad_wide(xs, gs, nbr, tile8t, n, cin, cout, gw=gw, gb=gb, defer=deferred)
        got.append((gw, gb))
    for cin, cout, ws in [(16, 8, (8, 1)), (32, 16, (16, 1)), (16, 24, (1, 16))]:
        blocked = cout % 8 == 0 and ws[1] == 1
        xs = _to_blocks(torch.randn(n, cin, device='cuda'))
        g = torch.randn(n, cout, device='cuda')
        gs = _to_blocks(g) if blocked else [g]
        a = (torch.full((cin * cout,), float('nan'), device='cuda'), torch.full((cout,), float('nan'), device='cuda'))
        b = (torch.full((cin * cout,), float('nan'), device='cuda'), torch.full((cout,), float('nan'), device='cuda'))
        ops.linear_wgrad_wide(xs, cin, gs, cout, a[0], ws[0], ws[1], a[1], g_blocked=blocked)
        ops.linear_wgrad_wide(xs, cin, gs, cout, b[0], ws[0], ws[1], b[1], g_blocked=blocked, defer=deferred)
        want.append(a); got.append(b)
    assert len(deferred) == 7 and bool(torch.isnan(got[0][0]).all())
    ops.wide_reduce_many(deferred)
    assert deferred == []
    for (w0, w1), (g0, g1) in zip(want, got):
        assert torch.equal(w0, g0) and torch.equal(w1, g1)


def test_sum_many_adds_in_list_order(pkg):
    """linr_sum_many: dst (+)= src[0] + src[1] + ... with the sources added in list order - bitwise what chained additions give."""
    import ctypes
    from linr_pcgc_amd import _lib
    L = _lib.lib()
    torch.manual_seed(3)
    n = 4 * 12345
    srcs = [torch.randn(n, device='cuda') for _ in range(7)]
    dst = torch.randn(n, device='cuda')
    want = dst.clone()
    for t in srcs:
        want = want + t
    arr = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in srcs])
    _lib.check(L.linr_sum_many(arr, 7, n, dst.data_ptr(), 1, torch.cuda.current_stream().cuda_stream), 'linr_sum_many')
    assert torch.equal(dst, want)
    _lib.check(L.linr_sum_many(arr, 3, n, dst.data_ptr(), 0, torch.cuda.current_stream().cuda_stream), 'linr_sum_many')
    assert torch.equal(dst, (srcs[0] + srcs[1]) + srcs[2])


def test_wide_results_do_not_depend_on_leftover_onchip_state(pkg):
    """The wide executor's kernels may not read LDS or registers they did not write (tests/test_gpu_parity.py::
    test_results_do_not_depend_on_leftover_onchip_state for the 8-wide executor): with every entry of csrc/wide.hip preceded by the
    kernel that fills the LDS and vector registers of all CUs with 0xFFFFFFFF (linr_debug_poison bit 16; the scale context's kernels:
    bits 11, 12) three training steps at width 16 give the same parameters, moments and bits, bit for bit."""
    from linr_pcgc_amd import _lib, overfit, synthetic
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('sphere8', 0, 'cuda')], None, 64, 'cuda')
    L = _lib.lib()

    def run(mask):
        m = overfit.gen_model(gop.scale_num, 'cuda', seed=8807, hidden=16)
        o = FlatAdam(m)
        bits = torch.zeros(3, dtype=torch.float64, device='cuda')
        L.linr_debug_poison(mask)
        try:
            for s in range(3):
                train_step(m, o, gop.frames[0], gop.point_nums[0], out=bits[s:s + 1])
            torch.cuda.synchronize()
        finally:
            L.linr_debug_poison(0x1FFFF if os.environ.get('LINR_DEBUG_POISON') else 0)
        return m.flat_parameters().clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), bits.cpu()
    clean, dirty = run(0), run(0x1FFFF)
    assert bool(torch.isfinite(dirty[0]).all())
    for a, b in zip(clean, dirty):
        assert torch.equal(a, b)


@pytest.mark.parametrize('hidden,block_layers', [(16, 1), (32, 1), (16, 2), (32, 3)])
def test_wide_fused_pointwise_epilogues_equal_the_separate_launches(pkg, hidden, block_layers):
    """conv1_0 / conv1_2 of a wide Inception layer and their backward-data passes in the convolutions' epilogues (linr_spconv_wide_pw)
    keep the fmaf chains of the stand-alone pointwise kernel: two training steps give the same parameters, moments and bits, bit for
    bit, as with LINR_WIDE_FUSE_PW=0 (one launch per pointwise layer)."""
    from linr_pcgc_amd import overfit, synthetic, wide_net
    from linr_pcgc_amd.model_core import FlatAdam, train_step
    gop = overfit.Gop(None, [synthetic.sequence_frame_device('sphere8', 0, 'cuda')], None, 64, 'cuda', block_layers=block_layers)

    def run(fuse):
        old = wide_net._FUSE_PW
        wide_net._FUSE_PW = fuse
        try:
            m = overfit.gen_model(gop.scale_num, 'cuda', seed=8807, hidden=hidden, block_layers=block_layers)
            o = FlatAdam(m)
            bits = torch.zeros(2, dtype=torch.float64, device='cuda')
            for s in range(2):
                train_step(m, o, gop.frames[0], gop.point_nums[0], out=bits[s:s + 1])
            torch.cuda.synchronize()
        finally:
            wide_net._FUSE_PW = old
        return m.flat_parameters().clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), bits.cpu()
    a, b = run(True), run(False)
    assert bool(torch.isfinite(a[0]).all())
    for x, y in zip(a, b):
        assert torch.equal(x, y)
