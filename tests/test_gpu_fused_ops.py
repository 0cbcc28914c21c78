"""GPU parity of the executor's fused layers through their own C-ABI entries (include/linr_hip.h, "fused layers as
stand-alone ops") against the matching pieces of the CPU oracle (oracle/network.py):
  linr_sce_fwd / _bwd           models/model_core.py:48-53
  linr_head_fwd / _bwd          models/upsample.py:137-161 + models/model_core.py:76-81
  linr_inception_fwd / _bwd_data, linr_spconv_wgrad_dual44     models/resnet.py:55-60
  linr_occ_conv7                models/upsample.py:206-214 (first conv + ReLU of the 7 outter blocks)
Tolerances as in test_gpu_parity.py: activations |d| <= 1e-4 + 1e-4 |x|, bits rel 1e-5, gradients 1e-4 of the tensor's
own largest entry (these are single layers: no depth to amplify rounding).
"""
import ctypes
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import network as onet          # noqa: E402
from oracle import octree as ooct           # noqa: E402


@pytest.fixture(scope='module')
def env(golden_dir):
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    from linr_pcgc_amd import _lib, ops
    L = _lib.lib()
    g = np.load(os.path.join(golden_dir, 'octree_shell128.npz'))
    dev = torch.device('cuda:0')
    coord = g['s0_coord']
    nbr_np = ooct.neighbour_table(coord)
    nbr = ops.kmap_build(torch.from_numpy(coord).to(dev))
    lo, mask = ops.kmap_compress(nbr)
    return {'L': L, 'lib': _lib, 'dev': dev, 'n': len(coord), 'nbr': nbr, 'lo': lo, 'mask': mask, 'ld': nbr.shape[1],
            'nbr_t': torch.from_numpy(nbr_np).long(), 'g': g}


def _padded(t, dev):
    """[n, c] host tensor -> device matrix with the all-zero row at index -1 (LINR_PAD_ROW contract); returns (buf, view)."""
    buf = torch.zeros((t.shape[0] + 1, t.shape[1]), device=dev)
    buf[1:] = t.to(dev)
    return buf, buf[1:]


def _empty_padded(n, c, dev):
    buf = torch.zeros((n + 1, c), device=dev)
    return buf, buf[1:]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _close(a, b, rtol, atol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), '%s: max err %.3e' % (what, float(err.max()))


def _rel_own_max(a, b, what, rtol=1e-4):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    gmax = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= rtol * gmax + 1e-9, '%s: err %.3e, own max %.3e' % (what, err, gmax)


def _inc_params(gen, dev):
    shapes = {'w00': (27, 8, 4), 'b00': (4,), 'w01': (27, 4, 4), 'b01': (4,), 'w10': (8, 4), 'b10': (4,), 'w11': (27, 4, 4),
              'b11': (4,), 'w12': (4, 4), 'b12': (4,)}
    host = {k: (torch.randn(*s, generator=gen) * 0.2) for k, s in shapes.items()}
    devt = {k: v.to(dev).contiguous() for k, v in host.items()}
    return host, devt


def _inc_struct(env, devt):
    return env['lib'].LinrInceptionParams(**{k: v.data_ptr() for k, v in devt.items()})


def _oracle_inception(x, nbr, w):
    sd = {'p.conv0_0.kernel': w['w00'], 'p.conv0_0.bias': w['b00'].view(1, -1), 'p.conv0_1.kernel': w['w01'],
          'p.conv0_1.bias': w['b01'].view(1, -1), 'p.conv1_0.kernel': w['w10'], 'p.conv1_0.bias': w['b10'].view(1, -1),
          'p.conv1_1.kernel': w['w11'], 'p.conv1_1.bias': w['b11'].view(1, -1), 'p.conv1_2.kernel': w['w12'],
          'p.conv1_2.bias': w['b12'].view(1, -1)}
    return onet.inception(x, nbr, sd, 'p')


def test_inception_layer_forward_backward(env):
    """linr_inception_fwd + linr_inception_bwd_data + linr_spconv_wgrad_dual44 against autograd through
    oracle.network.inception (models/resnet.py:55-60), with and without the ReLU mask of the layer's input."""
    L, dev, n = env['L'], env['dev'], env['n']
    gen = torch.Generator().manual_seed(5)
    x_h = torch.relu(torch.randn(n, 8, generator=gen))               # the layer's input is a ReLU output in make_block
    w_h, w_d = _inc_params(gen, dev)
    q = _inc_struct(env, w_d)
    _, x = _padded(x_h, dev)
    _, H = _empty_padded(n, 8, dev)
    _, M = _empty_padded(n, 4, dev)
    _, I = _empty_padded(n, 8, dev)
    env['lib'].check(L.linr_inception_fwd(x.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n,
                                          ctypes.byref(q), H.data_ptr(), M.data_ptr(), I.data_ptr(), _stream()), 'linr_inception_fwd')
    xo = x_h.clone().requires_grad_()
    wo = {k: v.clone().requires_grad_() for k, v in w_h.items()}
    ref = _oracle_inception(xo, env['nbr_t'], wo)
    _close(I, ref, 1e-4, 1e-4, 'inception output')
    h_ref = torch.cat([F.relu(onet.conv3(x_h, env['nbr_t'], w_h['w00'], w_h['b00'].view(1, -1))),
                       F.relu(onet.conv1(x_h, w_h['w10'], w_h['b10'].view(1, -1)))], dim=1)
    _close(H, h_ref, 1e-4, 1e-4, 'H')
    gI_h = torch.randn(n, 8, generator=gen)
    ref.backward(gI_h)
    _, gI = _padded(gI_h, dev)
    _, gM = _empty_padded(n, 4, dev)
    _, gH = _empty_padded(n, 8, dev)
    _, gX = _empty_padded(n, 8, dev)
    args = (gI.data_ptr(), x.data_ptr(), H.data_ptr(), M.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n,
            ctypes.byref(q), gM.data_ptr(), gH.data_ptr(), gX.data_ptr())
    env['lib'].check(L.linr_inception_bwd_data(*args, 0, _stream()), 'linr_inception_bwd_data')
    _close(gX, xo.grad, 1e-4, 1e-4, 'input gradient')
    # LINR_RELU_MASK | LINR_ACCUM: (old + result) * (x > 0), the form block_in's layer 0 uses with the ResNetBlock skip
    old = torch.randn(n, 8, generator=gen)
    gX[:] = old.to(dev)
    env['lib'].check(L.linr_inception_bwd_data(*args, env['lib'].LINR_RELU_MASK | env['lib'].LINR_ACCUM, _stream()), 'bwd masked')
    _close(gX, (xo.grad + old) * (x_h > 0), 1e-4, 1e-4, 'masked accumulated input gradient')
    # weight gradients of the two 4->4 convolutions (one pass) and of the rest through the plain entries
    slab = torch.empty((512, 872), device=dev)
    env['lib'].check(L.linr_spconv_wgrad_dual44(H.data_ptr(), gI.data_ptr(), 8, gM.data_ptr(), 4, env['nbr'].data_ptr(), None,
                                                env['ld'], n, slab.data_ptr(), _stream()), 'linr_spconv_wgrad_dual44')
    tot = slab.double().sum(0).cpu()
    _rel_own_max(tot[:432].view(27, 4, 4), wo['w01'].grad, 'gW01')
    _rel_own_max(tot[432:436], wo['b01'].grad, 'gb01')
    _rel_own_max(tot[436:868].view(27, 4, 4), wo['w11'].grad, 'gW11')
    _rel_own_max(tot[868:872], wo['b11'].grad, 'gb11')
    # ... and with coalesced gathers + LDS transpose (linr_kmap_tile8t; needs ld % 4 == 0): the same partials
    from linr_pcgc_amd import ops
    ld4 = (n + 63) // 64 * 64
    nbr4 = torch.full((27, ld4), -1, dtype=torch.int32, device=dev)
    nbr4[:, :n] = env['nbr']
    t8t = ops.kmap_tile8t(nbr4, n)
    slab4 = torch.empty_like(slab)
    env['lib'].check(L.linr_spconv_wgrad_dual44(H.data_ptr(), gI.data_ptr(), 8, gM.data_ptr(), 4, nbr4.data_ptr(), t8t.data_ptr(),
                                                ld4, n, slab4.data_ptr(), _stream()), 'wgrad_dual44 transposing')
    assert torch.equal(slab, slab4), 'the transposing kernel must give the same partials'


def test_head_forward_backward(env):
    """linr_head_fwd / linr_head_bwd against basic_module + BCE of the oracle (upsample.py:137-161, model_core.py:76-81)."""
    L, dev, n = env['L'], env['dev'], env['n']
    gen = torch.Generator().manual_seed(11)
    prior_h = torch.randn(n, 8, generator=gen)
    Wp = (torch.randn(27, 8, 8, generator=gen) * 0.15)
    bp = torch.randn(8, generator=gen) * 0.1
    w1 = torch.randn(24, 8, generator=gen) * 0.5
    b1 = torch.randn(24, generator=gen) * 0.1
    w2 = torch.randn(1, 24, generator=gen) * 0.5
    b2 = torch.randn(1, generator=gen) * 0.1
    occ_h = torch.from_numpy(env['g']['s0_occ']).float()
    k = 3
    d = lambda t: t.to(dev).contiguous()
    _, prior = _padded(prior_h, dev)
    occ = d(occ_h)
    c_out = torch.empty((n, 8), device=dev)
    p_out = torch.empty(n, device=dev)
    bits = torch.zeros(1, dtype=torch.float64, device=dev)
    ws = torch.empty(L.linr_head_workspace_bytes(n), dtype=torch.uint8, device=dev)
    Wd, bd, w1d, b1d, w2d, b2d = d(Wp), d(bp), d(w1), d(b1), d(w2), d(b2)
    env['lib'].check(L.linr_head_fwd(prior.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n, Wd.data_ptr(),
                                     bd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(),
                                     occ.data_ptr() + 4 * k, 8, c_out.data_ptr(), p_out.data_ptr(), bits.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _stream()), 'linr_head_fwd')
    leaves = [t.clone().requires_grad_() for t in (prior_h, Wp, bp, w1, b1, w2, b2)]
    po, Wo, bo, w1o, b1o, w2o, b2o = leaves
    c_ref = onet.conv3(po, env['nbr_t'], Wo, bo.view(1, -1))
    c_ref.retain_grad()
    z = F.linear(F.relu(F.linear(c_ref, w1o, b1o)), w2o, b2o)
    p_ref = torch.sigmoid(z)
    bits_ref = F.binary_cross_entropy(p_ref, occ_h[:, k:k + 1], reduction='sum') / math.log(2.0)
    _close(c_out, c_ref, 1e-4, 1e-4, 'C')
    _close(p_out, p_ref.view(-1), 1e-4, 1e-6, 'p')
    assert abs(float(bits) - float(bits_ref)) <= 1e-5 * float(bits_ref)
    # decoder form: no target, no bits - same probabilities, bit for bit
    p2 = torch.empty_like(p_out)
    env['lib'].check(L.linr_head_fwd(prior.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n, Wd.data_ptr(),
                                     bd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), None, 0,
                                     c_out.data_ptr(), p2.data_ptr(), None, None, 0, _stream()), 'linr_head_fwd (decoder)')
    assert torch.equal(p_out, p2)
    gscale = 1.0 / 12345.0
    (bits_ref * gscale).backward()
    gc = torch.empty((n, 8), device=dev)
    ghead = torch.empty(241, device=dev)
    env['lib'].check(L.linr_head_bwd(c_out.data_ptr(), p_out.data_ptr(), occ.data_ptr() + 4 * k, 8, w1d.data_ptr(), b1d.data_ptr(),
                                     w2d.data_ptr(), gscale, gc.data_ptr(), n, ghead.data_ptr(), ws.data_ptr(), ws.numel(),
                                     _stream()), 'linr_head_bwd')
    _rel_own_max(gc, c_ref.grad, 'gC')
    gh = ghead.cpu()
    _rel_own_max(gh[:192].view(24, 8), w1o.grad, 'gW1')
    _rel_own_max(gh[192:216], b1o.grad, 'gb1')
    _rel_own_max(gh[216:240].view(1, 24), w2o.grad, 'gw2')
    _rel_own_max(gh[240:241], b2o.grad, 'gb2')
    g2 = torch.empty_like(ghead)
    env['lib'].check(L.linr_head_bwd(c_out.data_ptr(), p_out.data_ptr(), occ.data_ptr() + 4 * k, 8, w1d.data_ptr(), b1d.data_ptr(),
                                     w2d.data_ptr(), gscale, gc.data_ptr(), n, g2.data_ptr(), ws.data_ptr(), ws.numel(),
                                     _stream()), 'linr_head_bwd again')
    assert torch.equal(ghead, g2), 'deterministic reduction'


def test_occ_conv7_matches_seven_oracle_convs(env):
    """linr_occ_conv7: relu(conv3(occ[:, :g+1]; W_g) + b_g) for the 7 outter blocks from ONE gather (upsample.py:206-214)."""
    L, dev, n = env['L'], env['dev'], env['n']
    gen = torch.Generator().manual_seed(3)
    occ_h = torch.from_numpy(env['g']['s0_occ']).float()
    _, occ = _padded(occ_h, dev)
    ws_h, bs_h, w_off, b_off, cur = [], [], [], [], 5            # an arbitrary non-zero base offset inside `params`
    chunks = [torch.zeros(5)]
    for g in range(7):
        w = torch.randn(27, g + 1, 8, generator=gen) * 0.3
        b = torch.randn(8, generator=gen) * 0.1
        ws_h.append(w); bs_h.append(b)
        w_off.append(cur); chunks.append(w.reshape(-1)); cur += w.numel()
        b_off.append(cur); chunks.append(b); cur += 8
    params = torch.cat(chunks).to(dev)
    out = torch.zeros((7, n + 1, 8), device=dev)
    o_off = [g * (n + 1) * 8 for g in range(7)]
    arr = lambda v: (ctypes.c_int64 * 7)(*v)
    env['lib'].check(L.linr_occ_conv7(occ.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n, params.data_ptr(),
                                      arr(w_off), arr(b_off), out[0, 1:].data_ptr(), arr(o_off), _stream()), 'linr_occ_conv7')
    for g in range(7):
        ref = F.relu(onet.conv3(occ_h[:, :g + 1], env['nbr_t'], ws_h[g], bs_h[g].view(1, -1)))
        _close(out[g, 1:], ref, 1e-4, 1e-4, 'first conv of outter block %d' % (g + 1))


def test_scale_context_forward_backward(env, golden_dir):
    """linr_sce_fwd / linr_sce_bwd over a 3-scale frame against oracle.scale_context (model_core.py:48-53) + autograd."""
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    L, dev, g = env['L'], env['dev'], env['g']
    torch.manual_seed(21)
    model = LINR_PCGC_Model({'scale_num': 5, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 2, 'outstage': 8, 'instage': 1})
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    use = [(0, 2), (1, 0), (2, 4)]                             # (scale of the cloud, scale_idx of the model) - not in order
    offs = [torch.from_numpy(g['s%d_offset' % s]).float() for s, _ in use]
    rows = [o.shape[0] for o in offs]
    row_off = np.zeros(4, dtype=np.int64); row_off[1:] = np.cumsum(rows)
    sidx = np.asarray([si for _, si in use], dtype=np.int32)
    R = int(row_off[-1])
    off_feat = torch.cat(offs).to(dev).contiguous()
    fr = env['lib'].LinrFrame(rows=R, n_scales=3, model_scale_num=5, block_layers=2, flags=0, row_off_h=row_off.ctypes.data,
                              scale_idx_h=sidx.ctypes.data, nbr=0, nbr_ld=R, nbr_lo=0, nbr_mask=0, offset_feat=off_feat.data_ptr(), occ=0, nbr8t=0)
    mix, hid, x0 = (torch.empty((R, c), device=dev) for c in (16, 16, 8))
    env['lib'].check(L.linr_sce_fwd(model.flat_parameters().data_ptr(), ctypes.byref(fr), mix.data_ptr(), hid.data_ptr(), x0.data_ptr(),
                                    _stream()), 'linr_sce_fwd')
    gx0_h = torch.randn(R, 8, generator=torch.Generator().manual_seed(2))
    gx0 = gx0_h.to(dev)
    ghid = torch.empty((R, 16), device=dev)
    env['lib'].check(L.linr_sce_bwd(model.flat_parameters().data_ptr(), ctypes.byref(fr), gx0.data_ptr(), hid.data_ptr(), ghid.data_ptr(),
                                    _stream()), 'linr_sce_bwd')
    for j, (s, si) in enumerate(use):
        a, b = int(row_off[j]), int(row_off[j + 1])
        ref = onet.scale_context(sd, offs[j], si)
        _close(x0[a:b], ref, 1e-4, 1e-5, 'x_low of scale_idx %d' % si)
        h_ref = F.relu(F.linear(torch.cat([sd['scale_emb.weight'][si].expand(b - a, -1), offs[j]], 1), sd['scale_mlp.%d.0.weight' % si],
                                sd['scale_mlp.%d.0.bias' % si]))
        _close(hid[a:b], h_ref, 1e-4, 1e-5, 'hidden layer')
        gh_ref = (gx0_h[a:b] @ sd['scale_mlp.%d.2.weight' % si]) * (h_ref > 0)
        _close(ghid[a:b], gh_ref, 1e-4, 1e-5, 'hidden gradient')
        assert torch.equal(mix[a:b, 8:15].cpu(), offs[j]) and bool((mix[a:b, 15] == 0).all())
    # the complete backward (linr_sce_bwd_params): gradients of scale_emb and the three scales' MLPs against autograd, zeros for the
    # scales the frame does not contain; mix is optional in the forward
    hid2, x02 = torch.empty_like(hid), torch.empty_like(x0)
    env['lib'].check(L.linr_sce_fwd(model.flat_parameters().data_ptr(), ctypes.byref(fr), None, hid2.data_ptr(), x02.data_ptr(), _stream()),
                     'linr_sce_fwd')
    assert torch.equal(hid2, hid) and torch.equal(x02, x0)
    npar = int(L.linr_sce_param_count(5))
    assert npar == 5 * 8 + 5 * (16 * 15 + 16 + 8 * 16 + 8)
    slab = env['lib'].scratch(L.linr_sce_bwd_params_slab_bytes(5), dev)
    grads = torch.full((npar,), float('nan'), device=dev)
    env['lib'].check(L.linr_sce_bwd_params(model.flat_parameters().data_ptr(), ctypes.byref(fr), gx0.data_ptr(), hid.data_ptr(),
                                           slab.data_ptr(), slab.numel(), grads.data_ptr(), _stream()), 'linr_sce_bwd_params')
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items() if k.startswith('scale_')}
    loss = sum((onet.scale_context(sdg, offs[j], si) * gx0_h[int(row_off[j]):int(row_off[j + 1])]).sum() for j, (_, si) in enumerate(use))
    loss.backward()
    names = ['scale_emb.weight'] + ['scale_mlp.%d.%d.%s' % (si, l, w) for si in range(5) for l in (0, 2) for w in ('weight', 'bias')]
    want = torch.cat([(sdg[k].grad if sdg[k].grad is not None else torch.zeros_like(sdg[k])).reshape(-1) for k in names])
    assert want.numel() == npar
    _close(grads, want, 1e-4, 1e-4 * float(want.abs().max()), 'scale-context parameter gradients')
    for si in (1, 3):                      # absent scales: exact zeros
        a0 = 40 + si * 392
        assert bool((grads[a0:a0 + 392] == 0).all()) and bool((grads[8 * si:8 * si + 8] == 0).all())
    # a frame descriptor that names a scale the model does not have is rejected
    bad = np.asarray([0, 1, 5], dtype=np.int32)
    fr.scale_idx_h = bad.ctypes.data
    assert L.linr_sce_fwd(model.flat_parameters().data_ptr(), ctypes.byref(fr), mix.data_ptr(), hid.data_ptr(), x0.data_ptr(), _stream()) == -1


@pytest.mark.parametrize('nblocks', [256, 4])
def test_conv88_backward_fused_one_gather(env, nblocks):
    """linr_spconv_bwd_fused (backward-data + weight gradient of a conv 8->8 from ONE gather of the output gradient): the input
    gradient is bit-identical to linr_spconv_cmap's backward-data, kernel / bias gradients match autograd of the oracle
    convolution (MinkowskiConvolution backward, a18) and every slab row is written."""
    from linr_pcgc_amd import ops
    dev, n = env['dev'], env['n']
    gen = torch.Generator().manual_seed(77 + nblocks)
    x_h = torch.randn(n, 8, generator=gen)
    go_h = torch.randn(n, 8, generator=gen)
    w_h = torch.randn(27, 8, 8, generator=gen) * 0.2
    xo = x_h.clone().requires_grad_()
    wo = w_h.clone().requires_grad_()
    bo = torch.zeros(1, 8, requires_grad=True)
    onet.conv3(xo, env['nbr_t'], wo, bo).backward(go_h)
    _, go = _padded(go_h, dev)
    x = x_h.to(dev).contiguous()
    w = w_h.to(dev).contiguous()
    ref_gin = ops.spconv_cmap(go, env['lo'], env['mask'], n, w, None, bwd=True)
    gin, slab = ops.spconv_bwd_fused(go, x, env['lo'], env['mask'], n, w, nblocks=nblocks, reduce=False)
    assert torch.equal(gin, ref_gin)
    assert bool(torch.isfinite(slab).all()), 'slab rows left unwritten'
    tot = slab.double().sum(dim=0)
    _rel_own_max(tot[:1728].view(27, 8, 8), wo.grad, 'fused kernel gradient')
    _rel_own_max(tot[1728:], bo.grad.reshape(-1), 'fused bias gradient')
    _close(gin, xo.grad, 1e-4, 1e-4, 'fused input gradient')
    # run-to-run reproducible (fixed fold order)
    gin2, slab2 = ops.spconv_bwd_fused(go, x, env['lo'], env['mask'], n, w, nblocks=nblocks, reduce=False)
    assert torch.equal(slab, slab2) and torch.equal(gin, gin2)


@pytest.mark.parametrize('nblocks', [256, 2])
def test_outter_first_conv_weight_gradients_one_gather(env, nblocks):
    """linr_occ_wgrad7 (weight gradients of the first convolutions of the 7 outter blocks, conv3(occ[:, :b] -> 8) for b = 1..7,
    from ONE gather of the occupancy rows; models/upsample.py:206-214): kernel and bias gradients match autograd of seven oracle
    convolutions (MinkowskiConvolution backward-weight, a18); only the rows the kernel reports are written; reproducible."""
    from linr_pcgc_amd import ops
    dev, n = env['dev'], env['n']
    gen = torch.Generator().manual_seed(911 + nblocks)
    occ_h = (torch.rand(n, 8, generator=gen) < 0.4).float()
    gos_h = [torch.randn(n, 8, generator=gen) for _ in range(7)]
    refs = []
    for b in range(1, 8):
        w = (torch.randn(27, b, 8, generator=gen) * 0.2).requires_grad_()
        bias = torch.zeros(1, 8, requires_grad=True)
        onet.conv3(occ_h[:, :b].contiguous(), env['nbr_t'], w, bias).backward(gos_h[b - 1])
        refs.append((w.grad, bias.grad.reshape(-1)))
    _, occ = _padded(occ_h, dev)
    gos = [g.to(dev).contiguous() for g in gos_h]
    gw, gb = ops.occ_wgrad7(occ, gos, env['lo'], env['mask'], n, nblocks=nblocks)
    for b in range(7):
        _rel_own_max(gw[b], refs[b][0], 'kernel gradient of block %d' % (b + 1))
        _rel_own_max(gb[b], refs[b][1], 'bias gradient of block %d' % (b + 1))
    gw2, gb2 = ops.occ_wgrad7(occ, gos, env['lo'], env['mask'], n, nblocks=nblocks)
    assert all(torch.equal(a, b) for a, b in zip(gw + gb, gw2 + gb2))


@pytest.mark.parametrize('nblocks,flags', [(256, 0), (3, 6)])
def test_inception_backward_fused_pairs(env, nblocks, flags):
    """linr_inception_bwd_fused (both conv pairs of an Inception layer's backward, each from ONE gather): gH and gX are
    bit-identical to linr_inception_bwd_data's, the kernel / bias gradients of conv0_0, conv0_1, conv1_1 and conv1_0 match autograd through
    oracle.network.inception (models/resnet.py:55-60); flags 6 = LINR_ACCUM | LINR_RELU_MASK (block_in's layer 0)."""
    L, dev, n = env['L'], env['dev'], env['n']
    gen = torch.Generator().manual_seed(23 + nblocks)
    x_h = torch.relu(torch.randn(n, 8, generator=gen))
    w_h, w_d = _inc_params(gen, dev)
    q = _inc_struct(env, w_d)
    _, x = _padded(x_h, dev)
    _, H = _empty_padded(n, 8, dev)
    _, M = _empty_padded(n, 4, dev)
    _, I = _empty_padded(n, 8, dev)
    env['lib'].check(L.linr_inception_fwd(x.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n,
                                          ctypes.byref(q), H.data_ptr(), M.data_ptr(), I.data_ptr(), _stream()), 'linr_inception_fwd')
    xo = x_h.clone().requires_grad_()
    wo = {k: v.clone().requires_grad_() for k, v in w_h.items()}
    ref = _oracle_inception(xo, env['nbr_t'], wo)
    gI_h = torch.randn(n, 8, generator=gen)
    ref.backward(gI_h)
    _, gI = _padded(gI_h, dev)
    _, gM = _empty_padded(n, 4, dev)
    _, gH = _empty_padded(n, 8, dev)
    _, gX = _empty_padded(n, 8, dev)
    old = torch.randn(n, 8, generator=gen).to(dev)
    gX[:] = old
    args = (gI.data_ptr(), x.data_ptr(), H.data_ptr(), M.data_ptr(), env['lo'].data_ptr(), env['mask'].data_ptr(), env['ld'], n,
            ctypes.byref(q), gM.data_ptr(), gH.data_ptr(), gX.data_ptr())
    env['lib'].check(L.linr_inception_bwd_data(*args, flags, _stream()), 'linr_inception_bwd_data')     # also leaves gM
    _, gH2 = _empty_padded(n, 8, dev)
    _, gX2 = _empty_padded(n, 8, dev)
    gX2[:] = old
    slab = torch.full((nblocks, 1776), float('nan'), device=dev)
    env['lib'].check(L.linr_inception_bwd_fused(gI.data_ptr(), gM.data_ptr(), x.data_ptr(), H.data_ptr(), env['lo'].data_ptr(),
                                                env['mask'].data_ptr(), env['ld'], n, ctypes.byref(q), gH2.data_ptr(), gX2.data_ptr(),
                                                flags, slab.data_ptr(), nblocks, _stream()), 'linr_inception_bwd_fused')
    assert torch.equal(gH, gH2) and torch.equal(gX, gX2)
    assert bool(torch.isfinite(slab).all()), 'slab rows left unwritten'
    tot = slab.double().sum(0).cpu()
    _rel_own_max(tot[:864].view(27, 8, 4), wo['w00'].grad, 'gW00')
    _rel_own_max(tot[864:868], wo['b00'].grad, 'gb00')
    _rel_own_max(tot[868:1300].view(27, 4, 4), wo['w01'].grad, 'gW01')
    _rel_own_max(tot[1300:1304], wo['b01'].grad, 'gb01')
    _rel_own_max(tot[1304:1736].view(27, 4, 4), wo['w11'].grad, 'gW11')
    _rel_own_max(tot[1736:1740], wo['b11'].grad, 'gb11')
    _rel_own_max(tot[1740:1772].view(8, 4), wo['w10'].grad, 'gW10 (conv1_0 rides in the conv0_0 launch)')
    _rel_own_max(tot[1772:1776], wo['b10'].grad, 'gb10')


class _single_stream:
    """LINR_FUSED_SPLIT=0 for the launches inside: conv_bwd_wgrad_single_k (one stream) instead of the wave-specialised conv_bwd_wgrad_k."""
    def __enter__(self):
        self.old = os.environ.get('LINR_FUSED_SPLIT')
        os.environ['LINR_FUSED_SPLIT'] = '0'

    def __exit__(self, *exc):
        if self.old is None:
            del os.environ['LINR_FUSED_SPLIT']
        else:
            os.environ['LINR_FUSED_SPLIT'] = self.old


@pytest.mark.parametrize('nblocks', [256, 5, 1])
def test_wave_specialised_fused_backward_is_bit_identical(env, nblocks):
    """conv_bwd_wgrad_k (producer / consumer wave pairs, csrc/fused_bwd_split.h) against conv_bwd_wgrad_single_k (one stream) on the
    same inputs: every input gradient and every slab partial bit-identical, for the conv 8->8 and for both conv pairs of an Inception
    layer (nblocks 5 / 1: waves without a tile only run the barriers; many tiles per wave)."""
    from linr_pcgc_amd import ops
    L, dev, n = env['L'], env['dev'], env['n']
    gen = torch.Generator().manual_seed(5 + nblocks)
    x_h = torch.relu(torch.randn(n, 8, generator=gen))
    go_h = torch.randn(n, 8, generator=gen)
    w = (torch.randn(27, 8, 8, generator=gen) * 0.2).to(dev).contiguous()
    _, go = _padded(go_h, dev)
    x = x_h.to(dev).contiguous()
    gin, slab = ops.spconv_bwd_fused(go, x, env['lo'], env['mask'], n, w, nblocks=nblocks, reduce=False)
    with _single_stream():
        gin1, slab1 = ops.spconv_bwd_fused(go, x, env['lo'], env['mask'], n, w, nblocks=nblocks, reduce=False)
    assert torch.equal(gin, gin1) and torch.equal(slab, slab1)
    w_h, w_d = _inc_params(gen, dev)
    q = _inc_struct(env, w_d)
    _, xp = _padded(x_h, dev)
    _, H = _padded(torch.relu(torch.randn(n, 8, generator=gen)), dev)
    _, gI = _padded(torch.randn(n, 8, generator=gen), dev)
    _, gM = _padded(torch.randn(n, 4, generator=gen), dev)
    old = torch.randn(n, 8, generator=gen).to(dev)
    outs = []
    for single in (False, True):
        _, gH = _empty_padded(n, 8, dev)
        _, gX = _empty_padded(n, 8, dev)
        gX[:] = old
        sl = torch.full((nblocks, 1776), float('nan'), device=dev)

        def run():
            env['lib'].check(L.linr_inception_bwd_fused(gI.data_ptr(), gM.data_ptr(), xp.data_ptr(), H.data_ptr(), env['lo'].data_ptr(),
                                                        env['mask'].data_ptr(), env['ld'], n, ctypes.byref(q), gH.data_ptr(), gX.data_ptr(),
                                                        6, sl.data_ptr(), nblocks, _stream()), 'linr_inception_bwd_fused')
        if single:
            with _single_stream():
                run()
        else:
            run()
        torch.cuda.synchronize()
        outs.append((gH.clone(), gX.clone(), sl.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_wave_specialised_backward_in_the_executor(pkg, shell):
    """The training step's gradients (grouped launches, the tail convolution's gM / conv1_2 epilogue) with the wave-specialised fused
    backward kernels and with the single-stream ones: bit-identical, for --block_layers 1 and 2."""
    from linr_pcgc_amd import engine
    from gpu_common import _model_and_oracle
    for block_layers in (1, 2):
        model, _ = _model_and_oracle(pkg, 5, block_layers=block_layers)
        frame = model.make_frame(shell['scales'])
        flat = model.flat_parameters()
        res = []
        for single in (False, True):
            bits = torch.zeros(1, dtype=torch.float64, device='cuda')
            grads = torch.zeros_like(flat)

            def run():
                engine.net_forward(frame, flat, 0, 8, None, bits)
                engine.net_backward(frame, flat, grads, 1.0 / shell['point_num'])
                torch.cuda.synchronize()
            if single:
                with _single_stream():
                    run()
            else:
                run()
            res.append(grads)
        assert float(res[0].abs().max()) > 0 and torch.equal(res[0], res[1])
