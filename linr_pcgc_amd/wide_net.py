"""Executor for ``hidden_channel_conv`` = 16 / 32 (main.py:520; models/upsample.py:38-76 ``channels=``, models/resnet.py:12-51
``channels // 2``): the same network with wider feature rows, on CHANNEL-BLOCKED activations.

Every C-wide activation is kept as C / 8 separate [rows + 1][8] matrices (zero row in front, what the kernels gather from).  Every
layer is ONE launch of a kernel of csrc/wide.hip: a convolution Ci -> Co ``linr_spconv_wide`` (every input block of a row gathered
once per tap, the weights of a tap as an A-operand image in LDS, all output channels from that one gather; the same for
backward-data at the mirrored taps; the Inception layer's pointwise convolutions conv1_0 / conv1_2 and their backward-data passes ride
in its epilogue: ``linr_spconv_wide_pw``), its weight gradient ``linr_spconv_wgrad_wide`` (one gather per input block for all gradient
blocks; conv0_1 and conv1_1 together: ``linr_spconv_wgrad_wide2``), the pointwise weight gradients ``linr_linear_wgrad_wide``, a stage's
head ``linr_head_wide_fwd`` and the backward of all 8 heads ``linr_head_wide_bwd``; the scale context (width-independent) runs on the
8-wide network's kernels (``linr_sce_fwd``, ``linr_sce_bwd_params``).  The ~64 slab reductions of a backward pass are deferred into
two launches (``linr_wide_reduce_many``).  Concatenations are free; the padded block buffers of a training step come from a pool.
Deterministic (fixed launch order, no atomics); the forward is the same launches in training, encoding and stage-by-stage decoding,
so streams decode losslessly.

The schedule is Python (~190 launches per step; GPU-bound on the BASELINE-sized frames, launch-bound on small ones): 5.9 / 18.1 ms per
training step at widths 16 / 32 on the loot-like frame (profiles/r04_wide.txt; ~3.7x / ~14x the convolution work of width 8).  The
8-wide model (every BASELINE config, the reference's default and its shipped checkpoint) never comes here.
"""
import os
import threading

import torch

from . import _lib, ops
from ._lib import LINR_ACCUM, LINR_NO_BIAS, LINR_RELU, LINR_RELU_MASK, check

B = 8


def _stream():
    return _lib.current_stream_handle()


class _Pool:
    """Zero-padded block buffers reused from step to step (train_step's schedule asks for the same sequence every time): the kernels
    never write a block's row 0, so it is cleared once when the buffer is made instead of once per request (88 fill launches per step at
    width 16).  A buffer is [nb][cap + 1][8]; a request for fewer rows gets views of its first rows."""

    def __init__(self):
        self.bufs, self.i = [], 0

    def reset(self):
        self.i = 0

    def take(self, n, nb, dev):
        i = self.i
        self.i += 1
        if i < len(self.bufs):
            b = self.bufs[i]
            want = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == 'cuda' else None)
            if b.shape[0] == nb and b.shape[1] > n and b.device.type == dev.type and want == b.device.index:
                return [b[j, 1:n + 1] for j in range(nb)]
        b = torch.empty((nb, n + 1, B), dtype=torch.float32, device=dev)
        b[:, 0].zero_()
        if i < len(self.bufs):
            self.bufs[i] = b
        else:
            self.bufs.append(b)
        return [b[j, 1:] for j in range(nb)]


_FUSE_PW = os.environ.get('LINR_WIDE_FUSE_PW', '1') != '0'          # the pointwise layers of an Inception layer in the convolutions' epilogues
# Per-THREAD state of the running pass (two threads may train two wide models at once, each call sees only its own): `pool` = the
# buffer pool of the running forward / backward (WideNet.forward(pool=True)), else fresh buffers; `defer` = the running backward's
# list of deferred weight-gradient reductions (one launch per 32 at its end), else None.  One MODEL is used by one thread at a
# time: WideNet.forward / backward hold the model's lock (its pool and the frame bound by _bind() are per-model state).
_TLS = threading.local()


def _pool():
    return getattr(_TLS, 'pool', None)


def _defer():
    return getattr(_TLS, 'defer', None)


def _blocks(n, nb, dev):
    """nb zero-padded [n, 8] matrices (views buf[1:] of [n + 1, 8] buffers whose row 0 is zero)."""
    if _pool() is not None:
        return _pool().take(n, nb, dev if isinstance(dev, torch.device) else torch.device(dev))
    buf = torch.empty((nb, n + 1, B), dtype=torch.float32, device=dev)
    buf[:, 0].zero_()
    return [buf[i, 1:] for i in range(nb)]


def _lin_bwd_data(gout, w, w_off, ws_ci, ws_co, cin, cout, out, act=None, accumulate=False):
    n = gout.shape[0]
    flags = (LINR_ACCUM if accumulate else 0) | (LINR_RELU_MASK if act is not None else 0)
    check(_lib.lib().linr_linear_bwd_data(gout.data_ptr(), gout.stride(0), n, w.data_ptr() + 4 * w_off, ws_ci, ws_co, cin, cout,
                                          0 if act is None else act.data_ptr(), 0 if act is None else act.stride(0),
                                          out.data_ptr(), out.stride(0), flags, _stream()), 'linr_linear_bwd_data')


def _axpy(src, dst, accumulate=True):
    check(_lib.lib().linr_axpy(src.data_ptr(), src.numel(), dst.data_ptr(), 1 if accumulate else 0, _stream()), 'linr_axpy')


class _Conv:
    """A 3x3x3 convolution Ci -> Co on blocked activations.  xs: Ci / 8 blocks (or one block with Ci < 8 live channels)."""

    def __init__(self, mod):
        self.mod = mod
        self.ci, self.co = mod.kernel.shape[1], mod.kernel.shape[2]
        self.nbi, self.nbo = (self.ci + B - 1) // B, self.co // B

    def fwd(self, net, xs, relu=False, res=None, pw=None):
        """ONE launch (linr_spconv_wide): every input block of a row is gathered once per tap and feeds all output channels.
        pw: a pointwise layer fused into the epilogue (ops.spconv_wide)."""
        n = xs[0].shape[0]
        outs = _blocks(n, self.nbo, xs[0].device)
        ops.spconv_wide(xs[:self.nbi], net.lo, net.mask, n, self.mod.kernel, self.mod.bias.reshape(-1), res=res, relu=relu, outs=outs, pw=pw)
        return outs

    def bwd(self, net, xs, gouts, gins=None, act=None, need_input_grad=True, res=None, pw=None, wgrad=True):
        """Parameter gradients into .grad; input gradient (+ res, masked by act > 0: the ReLU that produced xs) accumulated into
        gins (list of (buffer, has_content)) or returned as fresh blocks."""
        n = gouts[0].shape[0]
        # weight gradients: all (input block, gradient block) pairs as groups of grouped launches of the transposing 8-wide kernel, ONE
        # fixed-order reduction straight into the parameter gradients (views of the flat gradient)
        if wgrad:
            ops.spconv_wgrad_wide(xs[:self.nbi], gouts[:self.nbo], net.nbr_full, net.tile8t, n, self.ci, self.co,
                                  gw=self.mod.kernel.grad, gb=self.mod.bias.grad.reshape(-1), defer=_defer())
        if not need_input_grad:
            return None
        fresh = gins is None
        if fresh:
            gins = [[g, False] for g in _blocks(n, self.nbi, gouts[0].device)]
        # backward-data in one launch: the output gradient's blocks gathered once per (mirrored) tap for all input channels; blocks that
        # already hold a gradient are accumulated into (all of them or none: the callers fill a list uniformly)
        acc = gins[0][1]
        assert all(g[1] == acc for g in gins)
        ops.spconv_wide(gouts[:self.nbo], net.lo, net.mask, n, self.mod.kernel, None, bwd=True, act=act, res=res,
                        outs=[g[0] for g in gins], accumulate=acc, pw=pw)
        for g in gins:
            g[1] = True
        return [g for g, _ in gins] if fresh else None


class _Pointwise:
    """A 1x1 convolution (ME layout [Cin][Cout]) or an nn.Linear (torch layout [Cout][Cin]) on blocked inputs as ONE launch
    (linr_linear_wide); the output is blocked when cout is a multiple of 8, else one dense [n, cout] matrix."""

    def __init__(self, weight, bias, cin, cout, layout, blocked_out=True):
        self.w, self.b, self.cin, self.cout, self.layout = weight, bias, cin, cout, layout
        self.nbi = cin // B
        self.blocked_out = blocked_out and cout % B == 0
        self.nbo = cout // B if self.blocked_out else 1
        self.cw = B if self.blocked_out else cout
        self.ws = (cout, 1) if layout == 'me' else (1, cin)          # element (ci, co) at ci ws[0] + co ws[1]

    def fwd(self, xs, relu=False, res=None, padded=True):
        n, dev = xs[0].shape[0], xs[0].device
        outs = _blocks(n, self.nbo, dev) if (self.blocked_out and padded) else \
            [torch.empty((n, self.cw), dtype=torch.float32, device=dev) for _ in range(self.nbo)]
        return ops.linear_wide(xs[:self.nbi], self.cin, self.w, self.ws[0], self.ws[1], self.b.reshape(-1), self.cout, outs,
                               out_blocked=self.blocked_out, res=res, relu=relu)

    def wgrad(self, xs, gouts):
        """Parameter gradients only (the backward-data pass rides in a convolution's epilogue)."""
        ops.linear_wgrad_wide(xs[:self.nbi], self.cin, gouts[:self.nbo], self.cout, self.w.grad, self.ws[0], self.ws[1],
                              self.b.grad.reshape(-1), g_blocked=self.blocked_out, defer=_defer())

    def bwd(self, xs, gouts, gins=None, act=None, need_input_grad=True):
        """Parameter gradients into .grad (one grouped launch + one reduction); the input gradient (masked by act > 0) accumulated
        into gins (list of [buffer, has_content]) or returned as fresh blocks."""
        n = gouts[0].shape[0]
        ops.linear_wgrad_wide(xs[:self.nbi], self.cin, gouts[:self.nbo], self.cout, self.w.grad, self.ws[0], self.ws[1],
                              self.b.grad.reshape(-1), g_blocked=self.blocked_out, defer=_defer())
        if not need_input_grad:
            return None
        fresh = gins is None
        if fresh:
            gins = [[g, False] for g in _blocks(n, self.nbi, gouts[0].device)]
        acc = gins[0][1]
        assert all(g[1] == acc for g in gins)
        # backward-data: the same kernel with the roles of cin / cout and the weight strides swapped
        ops.linear_wide(gouts[:self.nbo], self.cout, self.w, self.ws[1], self.ws[0], None, self.cin, [g[0] for g in gins],
                        in_blocked=self.blocked_out, act=act, accumulate=acc)
        for g in gins:
            g[1] = True
        return [g for g, _ in gins] if fresh else None


class _Block:
    """make_block (models/upsample.py:88-97): conv3 -> ReLU -> ResNetBlock (Inception layers) -> conv3."""

    def __init__(self, seq, C):
        self.C, self.h = C, C // 2
        self.first, self.tail = _Conv(seq[0]), _Conv(seq[3])
        self.layers = []
        for lay in seq[2].layers:
            h = self.h
            self.layers.append({'c00': _Conv(lay.conv0_0), 'c01': _Conv(lay.conv0_1),
                                'c10': _Pointwise(lay.conv1_0.kernel, lay.conv1_0.bias, C, h, 'me'), 'c11': _Conv(lay.conv1_1),
                                'c12': _Pointwise(lay.conv1_2.kernel, lay.conv1_2.bias, h, h, 'me')})

    def fwd(self, net, xs, res=None):
        nh = self.h // B
        a = self.first.fwd(net, xs, relu=True)
        tape = {'in': xs, 'a': a, 'layers': []}
        x = a
        n, dev = xs[0].shape[0], xs[0].device
        for q in self.layers:
            # conv1_0 rides in conv0_0's launch (it reads the row itself: the centre tap), conv1_2 + the residual in conv1_1's: the
            # fmaf chains of the stand-alone pointwise kernel, so the bits are those of the five-launch form (LINR_WIDE_FUSE_PW=0)
            if _FUSE_PW:
                h1 = _blocks(n, nh, dev)
                h0 = q['c00'].fwd(net, x, relu=True, pw=(1, q['c10'].w, q['c10'].b.reshape(-1), None, h1))
            else:
                h0 = q['c00'].fwd(net, x, relu=True)
                h1 = q['c10'].fwd(x, relu=True)
            i_lo = q['c01'].fwd(net, h0, res=x[:nh])
            if _FUSE_PW:
                i_hi = _blocks(n, nh, dev)
                m = q['c11'].fwd(net, h1, relu=True, pw=(2, q['c12'].w, q['c12'].b.reshape(-1), x[nh:], i_hi))
            else:
                m = q['c11'].fwd(net, h1, relu=True)
                i_hi = q['c12'].fwd(m, res=x[nh:])
            tape['layers'].append({'x': x, 'h0': h0, 'h1': h1, 'm': m})
            x = i_lo + i_hi
        if len(self.layers) > 1:           # ResNetBlock.forward: out += x (models/resnet.py:160-161)
            for blk, ab in zip(x, a):
                _axpy(ab, blk)
        tape['il'] = x
        return self.tail.fwd(net, x, res=res), tape

    def bwd(self, net, tape, g_out, need_input_grad):
        """g_out: gradient blocks of the block output.  Returns the input gradient blocks (or None)."""
        nh = self.h // B
        n, dev = g_out[0].shape[0], g_out[0].device
        nl = len(self.layers)
        fuse = _FUSE_PW
        g_m_last = None
        if fuse:
            # gM of the LAST Inception layer (backward of conv1_2 and of M's ReLU) rides in the tail convolution's backward-data launch
            lastq, lastt = self.layers[-1], tape['layers'][-1]
            g_m_last = _blocks(n, nh, dev)
            g_il = self.tail.bwd(net, tape['il'], g_out, pw=(3, lastq['c12'].w, None, lastt['m'], g_m_last))
        else:
            g_il = self.tail.bwd(net, tape['il'], g_out)
        g_a = None
        if len(self.layers) > 1:           # the extra skip: a receives g_il as well
            g_a = _blocks(n, self.C // B, dev)
            for s, d in zip(g_il, g_a):
                _axpy(s, d, accumulate=False)
        g_i = g_il
        for li, (q, t) in enumerate(zip(reversed(self.layers), reversed(tape['layers']))):
            x = t['x']
            if fuse and li == 0:
                g_m = g_m_last
                q['c12'].wgrad(t['m'], g_i[nh:])
            else:
                g_m = q['c12'].bwd(t['m'], g_i[nh:], act=t['m'])
            # the weight gradients of conv0_1 and conv1_1 (same shape, independent) as ONE launch: alone each is one wave per SIMD
            pair = fuse and _defer() is not None and net.tile8t is not None and n > 0
            g_h1 = q['c11'].bwd(net, t['h1'], g_m, act=t['h1'], wgrad=not pair)
            g_h0 = q['c01'].bwd(net, t['h0'], g_i[:nh], act=t['h0'], wgrad=not pair)
            if pair:
                c01, c11 = q['c01'].mod, q['c11'].mod
                ops.spconv_wgrad_wide2(t['h0'], g_i[:nh], c01.kernel.grad, c01.bias.grad.reshape(-1), t['h1'], g_m, c11.kernel.grad,
                                       c11.bias.grad.reshape(-1), net.tile8t, n, _defer())
            # the layer's input gradient: the residual's share g_i rides in conv0_0's backward-data epilogue (+ res), conv1_0's share
            # is added last - and with it, for the block's first layer of a one-layer block, the ReLU mask of a = relu(first conv)
            last_mask = tape['a'] if (nl == 1 and li == nl - 1) else None
            if fuse:
                q['c10'].wgrad(x, g_h1)
                g_x = q['c00'].bwd(net, x, g_h0, res=g_i, act=last_mask, pw=(4, q['c10'].w, None, g_h1, None))
            else:
                g_x = q['c00'].bwd(net, x, g_h0, res=g_i)
                q['c10'].bwd(x, g_h1, gins=[[g, True] for g in g_x], act=last_mask)
            g_i = g_x
        if g_a is not None:
            for s, d in zip(g_a, g_i):
                _axpy(s, d)
        if nl != 1:
            # g_i is the gradient of a = relu(first conv): mask it
            for g, ab in zip(g_i, tape['a']):
                _mask_inplace(g, ab)
        return self.first.bwd(net, tape['in'], g_i, need_input_grad=need_input_grad)


def _mask_inplace(g, act):
    """g *= (act > 0) through the pointwise kernel's ReLU-mask epilogue on an identity-free path: g = 0 * W + g masked."""
    # linr_linear_bwd_data with cin = cout = 8, a zero weight block, accumulate and mask: out = (0 + old) * (act > 0)
    z = _zeros_w(g.device)
    _lin_bwd_data(g, z, 0, 8, 1, B, B, g, act=act, accumulate=True)


_ZW = {}


def _zeros_w(dev):
    key = str(dev)
    if key not in _ZW:
        _ZW[key] = torch.zeros(64, dtype=torch.float32, device=dev)
    return _ZW[key]


class WideNet:
    def __init__(self, model, hidden):
        if hidden % 16 != 0 or hidden > 32:
            raise ValueError('hidden_channel_conv must be 8 (the tuned kernels) or 16 / 32 (channel-blocked executor), got %d' % hidden)
        self.model, self.C = model, hidden
        self._built = False
        self._pool = None
        # one thread at a time per model: the pool and the frame bound by _bind() are per-model state (train_step holds it across forward,
        # backward and the optimiser step)
        self.lock = threading.RLock()

    def _build(self):
        up = self.model.upsampler
        C = self.C
        self.block_in = _Block(up.block_in, C)
        self.outter = [_Block(b, C) for b in up.outter_blocks]
        self.prune = [_Conv(p[0].conv) for p in up.prune_blocks]
        self.heads = [(mlp[0][0], mlp[0][2]) for mlp in up.inner_mlps]          # PointwiseMLP([C, 24, 1]) = Linear, ReLU, Linear
        self._built = True

    # ---- shared pieces ------------------------------------------------------------------------------------------------
    def _bind(self, frame):
        if not self._built:
            self._build()
        # the op-level wrappers take the row count from the table's width: a view of its first `rows` columns (same stride)
        self.nbr, self.lo, self.mask = frame.nbr[:, :frame.rows], frame.nbr_lo, frame.nbr_mask
        self.nbr_full, self.tile8t = frame.nbr, frame.nbr8t
        self.n = frame.rows

    def _scale_context(self, frame, keep):
        """x0 [rows, 8] (one block): PointwiseMLP([15, 16, 8]) of the rows' scale on [emb | offset features] (model_core.py:48-53) - all
        scales in ONE launch of the library's scale-context kernel (linr_sce_fwd: the parameters of the scale context lead the flat
        parameter order whatever the width of the rest of the network); the hidden layer is kept for the backward pass."""
        x0 = _blocks(frame.rows, 1, frame.device)
        hid = torch.empty((frame.rows, 16), dtype=torch.float32, device=frame.device)
        check(_lib.lib().linr_sce_fwd(self.model.flat_parameters().data_ptr(), frame.cref(), None, hid.data_ptr(), x0[0].data_ptr(),
                                      _stream()), 'linr_sce_fwd')
        return x0, (hid if keep else None)

    def _occ_block(self, frame):
        return [frame.occ]                     # [rows, 8] view of a buffer with the zero row in front (engine.Frame)

    def _head(self, k, prior, frame, p, partial):
        """Stage k behind `prior`: the prune convolution, then the head MLP + sigmoid (+ the stage's bits partials) as one launch."""
        c = self.prune[k].fwd(self, prior)
        lin0, lin2 = self.heads[k]
        ops.head_wide_fwd(c, lin0.weight, lin0.bias, lin2.weight, lin2.bias, frame.occ[:, k] if partial is not None else None, p,
                          partial=partial)
        return c

    # ---- forward ---------------------------------------------------------------------------------------------------------
    def forward(self, frame, k0, k1, probs, bits, keep=False, pool=False):
        """Stages [k0, k1) teacher-forced on frame.occ (decoder: the columns decoded so far): probs [8, rows] rows k0..k1-1, bits
        (float64[1]) += their cost.  keep: record the activations for backward (k0 = 0, k1 = 8).  pool: the padded block buffers
        come from this executor's pool, i.e. they are valid until the NEXT pooled forward (train_step: forward, backward, done)."""
        with self.lock:
            self._bind(frame)
            if frame.rows == 0:
                return None
            if pool:
                if self._pool is None:
                    self._pool = _Pool()
                self._pool.reset()
            _TLS.pool = self._pool if pool else None
            try:
                return self._forward(frame, k0, k1, probs, bits, keep)
            finally:
                _TLS.pool = None

    def _forward(self, frame, k0, k1, probs, bits, keep):
        x0, sce_tape = self._scale_context(frame, keep)
        xg, bin_tape = self.block_in.fwd(self, x0)
        tape = {'sce': sce_tape, 'bin': bin_tape, 'xg': xg, 'stages': []} if keep else None
        nb = (frame.rows + 255) // 256                      # per-block bits partials of a stage (linr_head_wide_workspace_bytes / 8)
        parts = torch.empty(((k1 - k0) * nb,), dtype=torch.float64, device=frame.device) if bits is not None else None
        for k in range(k0, k1):
            blk_tape = None
            if k == 0:
                prior = xg
            else:
                occ = self._occ_block(frame)
                prior, blk_tape = self.outter[k - 1].fwd(self, occ, res=xg)
            p = probs[k] if probs is not None else torch.empty((frame.rows,), dtype=torch.float32, device=frame.device)
            c = self._head(k, prior, frame, p, None if parts is None else parts[(k - k0) * nb:(k - k0 + 1) * nb])
            if keep:
                tape['stages'].append({'prior': prior, 'c': c, 'p': p, 'blk': blk_tape})
        if parts is not None:
            ops.bits_finish(parts, (k1 - k0) * nb, bits)          # all stages' partials in one fixed-order pass
        return tape

    # ---- backward --------------------------------------------------------------------------------------------------------
    def backward(self, frame, tape, gscale, pool=False):
        """d (gscale * bits) / d params into the parameters' .grad (model._ensure_grad_views(): views of the flat gradient).
        pool: continue in the pool of the forward that made `tape`."""
        with self.lock:
            _TLS.pool = self._pool if pool else None
            _TLS.defer = []
            try:
                self._backward(frame, tape, gscale)
                ops.wide_reduce_many(_TLS.defer)          # the ~64 slab reductions of the pass, 32 per launch
            finally:
                _TLS.pool, _TLS.defer = None, None

    def _backward(self, frame, tape, gscale):
        self._bind(frame)
        self.model._ensure_grad_views()
        n, dev = frame.rows, frame.device
        C = self.C
        gz_scale = float(gscale) * 1.4426950408889634          # d(bits)/d(nats) = 1 / ln 2
        g_xg = [[g, False] for g in _blocks(n, C // B, dev)]
        # all 8 heads first, as one grouped launch (their inputs - c, p, the occupancy columns - exist once the forward has run): the
        # gradients of the prune convolutions' outputs and the heads' parameter gradients, straight into the flat gradient
        stages = tape['stages']
        g_cs = [_blocks(n, C // B, dev) for _ in range(8)]
        first = self.heads[0][0].weight.grad
        per = 24 * C + 49
        for k, (lin0, lin2) in enumerate(self.heads):           # the reference's parameter order: the 8 inner_mlps back to back
            assert lin0.weight.grad.data_ptr() == first.data_ptr() + 4 * k * per and lin2.bias.grad.data_ptr() == \
                first.data_ptr() + 4 * (k * per + per - 1)
        off0 = (first.data_ptr() - self.model._flat_grad.data_ptr()) // 4
        ops.head_wide_bwd([st['c'] for st in stages], [st['p'] for st in stages], [frame.occ[:, k] for k in range(8)],
                          [h[0].weight for h in self.heads], [h[0].bias for h in self.heads], [h[1].weight for h in self.heads],
                          gz_scale, g_cs, self.model._flat_grad[off0:off0 + 8 * per])
        g_priors = []
        for k in range(7, -1, -1):
            st = stages[k]
            g_c = g_cs[k]
            if k == 0:
                self.prune[k].bwd(self, st['prior'], g_c, gins=g_xg)
            else:
                g_prior = self.prune[k].bwd(self, st['prior'], g_c)
                g_priors.append(g_prior)                            # prior_k = x_glob + block output: both receive g_prior
                self.outter[k - 1].bwd(self, st['blk'], g_prior, need_input_grad=False)
        # x_glob's gradient: stage 0's (in g_xg) + the seven priors' - one pass per block instead of a read-modify-write pass per stage
        for b, (dst, _) in enumerate(g_xg):
            check(_lib.lib().linr_sum_many(ops._ptr_array([gp[b] for gp in g_priors]), len(g_priors), dst.numel(), dst.data_ptr(), 1,
                                           _stream()), 'linr_sum_many')
        g_x0 = self.block_in.bwd(self, tape['bin'], [g for g, _ in g_xg], need_input_grad=True)[0]
        # scale context: ghid, the four parameter gradients of every scale's MLP and the embedding rows in one call, straight into the
        # flat gradient (its leading linr_sce_param_count floats)
        L = _lib.lib()
        m = self.model
        slab = _lib.scratch(L.linr_sce_bwd_params_slab_bytes(m.scale_num), dev)
        check(L.linr_sce_bwd_params(m.flat_parameters().data_ptr(), frame.cref(), g_x0.data_ptr(), tape['sce'].data_ptr(), slab.data_ptr(),
                                    slab.numel(), m._flat_grad.data_ptr(), _stream()), 'linr_sce_bwd_params')
