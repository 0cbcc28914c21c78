"""Tensor-level wrappers of the op-level C-ABI entry points (include/linr_hip.h).

These are the MI355X replacements of the MinkowskiEngine operators the reference calls:
kernel-map build, 3x3x3 sparse convolution fwd / bwd-data / bwd-weight, 1x1 / nn.Linear layers, BCE bits, Adam.
Every function requires CUDA (ROCm) tensors and launches on the current PyTorch stream; nothing falls back to CPU.
"""
import torch

from . import _lib
from ._lib import LINR_ACCUM, LINR_NO_BIAS, LINR_PAD_ROW, LINR_RELU, LINR_RELU_MASK, check  # noqa: F401


def _stream():
    return _lib.current_stream_handle()


def _dev(t, dtype, name):
    if not t.is_cuda:
        raise _lib.LinrError('%s must be a GPU tensor: the coding network has no CPU path' % name)
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError('%s must be contiguous %s' % (name, dtype))
    return t


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def kmap_build(coords, validate=True):
    """coords int32 [N,3] sorted by the x-major ravel key (qscTensor order) -> nbr int32 [27,N] (-1 = absent)."""
    _dev(coords, torch.int32, 'coords')
    n = coords.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=coords.device)
    kmap_build_into(coords, nbr, 0, validate)
    return nbr


def kmap_build_into(coords, nbr, row_base, validate=True):
    """Fill columns [row_base, row_base+N) of a shared nbr [27, ld] with global row ids (multi-scale frames)."""
    L = _lib.lib()
    n = coords.shape[0]
    if n == 0:
        return
    if validate:
        bad = torch.zeros(1, dtype=torch.int32, device=coords.device)
        check(L.linr_kmap_validate(coords.data_ptr(), n, bad.data_ptr(), _stream()), 'linr_kmap_validate')
        if int(bad.item()) != 0:
            raise ValueError('coord must be unique, non-negative (< 2^20) and sorted by the x-major ravel key '
                             '(models/module_utils.py:246-256)')
    ws = _lib.scratch(L.linr_kmap_workspace_bytes(n), coords.device)
    check(L.linr_kmap_build(coords.data_ptr(), n, nbr.data_ptr(), nbr.shape[1], row_base, ws.data_ptr(), ws.numel(),
                            _stream()), 'linr_kmap_build')


def spconv_fwd(x, nbr, kernel, bias, res=None, relu=False, out=None, pad_row=False):
    """out = bias + sum_k x[nbr[k]] @ kernel[k] (+res)(ReLU).  x [N,ld>=cin] (a channel slice view is fine)."""
    n, cin, cout = nbr.shape[1], kernel.shape[1], kernel.shape[2]
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    flags = (LINR_RELU if relu else 0) | (LINR_PAD_ROW if pad_row else 0)
    check(_lib.lib().linr_spconv_fwd(x.data_ptr(), x.stride(0), nbr.data_ptr(), nbr.stride(0), n, kernel.data_ptr(),
                                     bias.data_ptr(), cin, cout, _ptr(res), 0 if res is None else res.stride(0),
                                     out.data_ptr(), out.stride(0), flags, _stream()), 'linr_spconv_fwd')
    return out


def spconv_bwd_data(gout, nbr, kernel, act=None, out=None, accumulate=False, pad_row=False):
    n, cin = nbr.shape[1], kernel.shape[1]
    cout = kernel.shape[2]
    if out is None:
        out = torch.empty((n, cin), dtype=torch.float32, device=gout.device)
    flags = (LINR_ACCUM if accumulate else 0) | (LINR_RELU_MASK if act is not None else 0) | \
            (LINR_PAD_ROW if pad_row else 0)
    check(_lib.lib().linr_spconv_bwd_data(gout.data_ptr(), gout.stride(0), nbr.data_ptr(), nbr.stride(0), n,
                                          kernel.data_ptr(), cin, cout, _ptr(act),
                                          0 if act is None else act.stride(0), out.data_ptr(), out.stride(0), flags,
                                          _stream()), 'linr_spconv_bwd_data')
    return out


def spconv_bwd_weight(x, gout, nbr, cin, cout, pad_row=False):
    """gW [27,cin,cout], gb [1,cout].  pad_row: x is a view buf[1:] of a buffer whose row 0 is zero (matrix-core kernel)."""
    n = nbr.shape[1]
    L = _lib.lib()
    gw = torch.empty((27, cin, cout), dtype=torch.float32, device=x.device)
    gb = torch.empty((1, cout), dtype=torch.float32, device=x.device)
    ws = _lib.scratch(max(L.linr_spconv_bwd_weight_workspace_bytes(n, cin, cout), 4), x.device)
    check(L.linr_spconv_bwd_weight(x.data_ptr(), x.stride(0), gout.data_ptr(), gout.stride(0), nbr.data_ptr(),
                                   nbr.stride(0), n, cin, cout, gw.data_ptr(), gb.data_ptr(), LINR_PAD_ROW if pad_row else 0,
                                   ws.data_ptr(),
                                   ws.numel(), _stream()), 'linr_spconv_bwd_weight')
    return gw, gb


def linear_fwd(x, weight, bias, cin, cout, layout, res=None, relu=False, out=None):
    """layout 'me' : weight [cin][cout] (MinkowskiConvolution kernel_size=1); 'torch': weight [cout][cin] (nn.Linear)."""
    n = x.shape[0]
    ws_ci, ws_co = (cout, 1) if layout == 'me' else (1, cin)
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    flags = (LINR_RELU if relu else 0) | (LINR_NO_BIAS if bias is None else 0)
    check(_lib.lib().linr_linear_fwd(x.data_ptr(), x.stride(0), n, weight.data_ptr(), ws_ci, ws_co, _ptr(bias), cin, cout,
                                     _ptr(res), 0 if res is None else res.stride(0), out.data_ptr(), out.stride(0),
                                     flags, _stream()), 'linr_linear_fwd')
    return out


def linear_bwd_data(gout, weight, cin, cout, layout, act=None, out=None, accumulate=False):
    n = gout.shape[0]
    ws_ci, ws_co = (cout, 1) if layout == 'me' else (1, cin)
    if out is None:
        out = torch.empty((n, cin), dtype=torch.float32, device=gout.device)
    flags = (LINR_ACCUM if accumulate else 0) | (LINR_RELU_MASK if act is not None else 0)
    check(_lib.lib().linr_linear_bwd_data(gout.data_ptr(), gout.stride(0), n, weight.data_ptr(), ws_ci, ws_co, cin, cout,
                                          _ptr(act), 0 if act is None else act.stride(0), out.data_ptr(),
                                          out.stride(0), flags, _stream()), 'linr_linear_bwd_data')
    return out


def linear_bwd_weight(x, gout, cin, cout, layout):
    n = x.shape[0]
    L = _lib.lib()
    ws_ci, ws_co = (cout, 1) if layout == 'me' else (1, cin)
    gw = torch.empty((cin, cout) if layout == 'me' else (cout, cin), dtype=torch.float32, device=x.device)
    gb = torch.empty((cout,), dtype=torch.float32, device=x.device)
    ws = _lib.scratch(max(L.linr_linear_bwd_weight_workspace_bytes(n, cin, cout), 4), x.device)
    check(L.linr_linear_bwd_weight(x.data_ptr(), x.stride(0), gout.data_ptr(), gout.stride(0), n, cin, cout,
                                   gw.data_ptr(), ws_ci, ws_co, gb.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                                   _stream()), 'linr_linear_bwd_weight')
    return gw, gb


def bce_bits_fwd(z, target):
    """z [N] logits, target [N] or a strided column view -> (p [N], bits double[1])."""
    n = z.shape[0]
    L = _lib.lib()
    p = torch.empty((n,), dtype=torch.float32, device=z.device)
    bits = torch.zeros(1, dtype=torch.float64, device=z.device)
    ws = _lib.scratch(max(L.linr_bce_workspace_bytes(n), 8), z.device)
    check(L.linr_bce_bits_fwd(z.data_ptr(), target.data_ptr(), target.stride(0), n, p.data_ptr(), bits.data_ptr(),
                              ws.data_ptr(), ws.numel(), _stream()), 'linr_bce_bits_fwd')
    return p, bits


def bce_bits_bwd(p, target, gscale):
    n = p.shape[0]
    gz = torch.empty((n,), dtype=torch.float32, device=p.device)
    check(_lib.lib().linr_bce_bits_bwd(p.data_ptr(), target.data_ptr(), target.stride(0), n, gscale, gz.data_ptr(),
                                       _stream()), 'linr_bce_bits_bwd')
    return gz


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """One fused torch.optim.Adam update over flat buffers (main.py:231-237,319); `step` is the 1-based step count."""
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    check(_lib.lib().linr_adam_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                    params.numel(), lr / bc1, bc2_sqrt, beta1, beta2, eps, weight_decay, _stream()),
          'linr_adam_step')


def kmap_compress(nbr, n=None):
    """nbr int32 [27, ld] (global row ids, x-major sorted rows) -> (lo int32 [9, ld], mask int32 [ld])."""
    ld = nbr.shape[1]
    n = ld if n is None else n
    lo = torch.zeros((9, ld), dtype=torch.int32, device=nbr.device)
    mask = torch.zeros((ld,), dtype=torch.int32, device=nbr.device)
    check(_lib.lib().linr_kmap_compress(nbr.data_ptr(), nbr.stride(0), n, lo.data_ptr(), mask.data_ptr(), ld, _stream()),
          'linr_kmap_compress')
    return lo, mask


def spconv_cmap(x, lo, mask, n, kernel, bias=None, bwd=False, res=None, act=None, relu=False, out=None, accumulate=False):
    """The executor's conv kernel (compressed map + MFMA).  x must be a view buf[1:] of a buffer whose row 0 is zero."""
    cin, cout = kernel.shape[1], kernel.shape[2]
    width = cin if bwd else cout
    if out is None:
        out = torch.empty((n, width), dtype=torch.float32, device=x.device)
    flags = (LINR_RELU if relu else 0) | (LINR_ACCUM if accumulate else 0) | (LINR_RELU_MASK if act is not None else 0) | \
        LINR_PAD_ROW
    check(_lib.lib().linr_spconv_cmap(1 if bwd else 0, x.data_ptr(), x.stride(0), lo.data_ptr(), mask.data_ptr(),
                                      lo.stride(0), n, kernel.data_ptr(), _ptr(bias), cin, cout, _ptr(res),
                                      0 if res is None else res.stride(0), _ptr(act), 0 if act is None else act.stride(0),
                                      out.data_ptr(), out.stride(0), flags, _stream()), 'linr_spconv_cmap')
    return out


def _ptr_array(blocks):
    import ctypes
    return (ctypes.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])


def spconv_wide(xs, lo, mask, n, kernel, bias=None, bwd=False, res=None, act=None, relu=False, outs=None, accumulate=False, pw=None):
    """linr_spconv_wide: a 3x3x3 convolution on channel-blocked activations, one gather of all input blocks per tap.
    xs: the gathered blocks (views buf[1:] of [n+1, 8] buffers whose row 0 is zero); kernel [27, cin, cout]; outs / res / act: lists
    of [n, 8] blocks of the produced side (cout / 8 forward, cin / 8 backward).  pw = (mode, W, b, aux blocks, out2 blocks): a pointwise
    layer of the wide Inception layer fused into the epilogue (linr_spconv_wide_pw, include/linr_hip.h)."""
    import ctypes
    cin, cout = kernel.shape[1], kernel.shape[2]
    npb = (cin if bwd else cout) // 8
    if outs is None:
        outs = [torch.empty((n, 8), dtype=torch.float32, device=xs[0].device) for _ in range(npb)]
    flags = (LINR_RELU if relu else 0) | (LINR_ACCUM if accumulate else 0) | (LINR_RELU_MASK if act is not None else 0)
    args = (1 if bwd else 0, _ptr_array(xs), lo.data_ptr(), mask.data_ptr(), lo.stride(0), n, kernel.data_ptr(), _ptr(bias), cin, cout,
            None if res is None else _ptr_array(res), None if act is None else _ptr_array(act), _ptr_array(outs), flags)
    if pw is None:
        check(_lib.lib().linr_spconv_wide(*args, _stream()), 'linr_spconv_wide')
    else:
        mode, w, b, aux, out2 = pw
        aux_a = None if aux is None else _ptr_array(aux)
        out_a = None if out2 is None else _ptr_array(out2)
        st = _lib.LinrWidePw(mode, w.data_ptr(), _ptr(b), None if aux_a is None else ctypes.cast(aux_a, ctypes.c_void_p),
                             None if out_a is None else ctypes.cast(out_a, ctypes.c_void_p))
        check(_lib.lib().linr_spconv_wide_pw(*args, ctypes.byref(st), _stream()), 'linr_spconv_wide_pw')
    return outs


def spconv_wgrad_wide(xs, gouts, nbr, tile8t, n, cin, cout, gw=None, gb=None, defer=None):
    """linr_spconv_wgrad_wide: kernel / bias gradient of a convolution on channel-blocked activations (one launch whose groups are the
    input blocks, one reduction).  gw [27, cin, cout] / gb [cout]: contiguous destinations (e.g. views of the flat gradient) or None.
    defer: a list - the partials stay in their slab and the reduction is appended to it for wide_reduce_many()."""
    L = _lib.lib()
    dev = xs[0].device
    if gw is None:
        gw = torch.empty((27, cin, cout), dtype=torch.float32, device=dev)
    if gb is None:
        gb = torch.empty((cout,), dtype=torch.float32, device=dev)
    assert gw.is_contiguous() and gb.is_contiguous()
    slab = _lib.scratch(L.linr_spconv_wgrad_wide_slab_bytes(cin, cout), dev)
    deferred = defer is not None and n > 0
    check(L.linr_spconv_wgrad_wide(_ptr_array(xs), cin, _ptr_array(gouts), cout, nbr.data_ptr(), _ptr(tile8t), nbr.stride(0), n,
                                   slab.data_ptr(), None if deferred else gw.data_ptr(), gb.data_ptr(), _stream()), 'linr_spconv_wgrad_wide')
    if deferred:
        nblk = int(L.linr_spconv_wgrad_wide_blocks(cout, 1 if tile8t is not None else 0))
        defer.append((_lib.LinrWideReduce(0, nblk, cin, cout, 0, 0, slab.data_ptr(), gw.data_ptr(), gb.data_ptr()), slab, gw, gb))
    return gw, gb


def spconv_wgrad_wide2(xsA, gsA, gwA, gbA, xsB, gsB, gwB, gbB, tile8t, n, defer):
    """linr_spconv_wgrad_wide2: the weight gradients of two h -> h convolutions (h = 8 len(xsA)) as one launch; their reductions are
    appended to `defer` (wide_reduce_many)."""
    L = _lib.lib()
    nb = len(xsA)
    h = 8 * nb
    stride = 2 * nb * nb * 1736
    slab = _lib.scratch(256 * stride * 4, xsA[0].device)
    check(L.linr_spconv_wgrad_wide2(_ptr_array(xsA), _ptr_array(gsA), _ptr_array(xsB), _ptr_array(gsB), h, tile8t.data_ptr(), n,
                                    slab.data_ptr(), _stream()), 'linr_spconv_wgrad_wide2')
    for cv, (gw, gb) in enumerate(((gwA, gbA), (gwB, gbB))):
        assert gw.is_contiguous() and gb.is_contiguous()
        defer.append((_lib.LinrWideReduce(0, 256, h, h, stride, 0, slab.data_ptr() + 4 * cv * nb * nb * 1736, gw.data_ptr(), gb.data_ptr()),
                      slab, gw, gb))


def wide_reduce_many(deferred):
    """linr_wide_reduce_many: the reductions collected by spconv_wgrad_wide / linear_wgrad_wide(defer=...), 32 per launch."""
    if not deferred:
        return
    arr = (_lib.LinrWideReduce * len(deferred))(*[d[0] for d in deferred])
    check(_lib.lib().linr_wide_reduce_many(arr, len(deferred), _stream()), 'linr_wide_reduce_many')
    deferred.clear()


def linear_wide(xs, cin, w, ws_ci, ws_co, bias, cout, outs, in_blocked=True, out_blocked=True, res=None, act=None, relu=False,
                accumulate=False):
    """linr_linear_wide: a pointwise layer on blocked activations as one launch.  xs / outs / res / act: lists of [n, 8] blocks, or
    (in_blocked / out_blocked False) one-element lists with a dense [n, channels] matrix.  Weight element (ci, co) at
    w.data_ptr() + 4 (ci ws_ci + co ws_co); backward-data: swap cin / cout and the strides."""
    n = xs[0].shape[0]
    flags = (LINR_RELU if relu else 0) | (LINR_ACCUM if accumulate else 0) | (LINR_RELU_MASK if act is not None else 0) | \
        (_lib.LINR_NO_BIAS if bias is None else 0)
    check(_lib.lib().linr_linear_wide(_ptr_array(xs), cin, 1 if in_blocked else 0, w.data_ptr(), ws_ci, ws_co, _ptr(bias), cout,
                                      1 if out_blocked else 0, None if res is None else _ptr_array(res),
                                      None if act is None else _ptr_array(act), _ptr_array(outs), n, flags, _stream()), 'linr_linear_wide')
    return outs


def linear_wgrad_wide(xs, cin, gouts, cout, gw, ws_ci, ws_co, gb, in_blocked=True, g_blocked=True, accumulate=False, defer=None):
    """linr_linear_wgrad_wide: gw(ci, co) (+)= sum_r x[r][ci] g[r][co] at gw.data_ptr() + 4 (ci ws_ci + co ws_co), gb[co] (+)= column sums.
    defer (a list, not with accumulate): the partials stay in the workspace, the reduction is appended for wide_reduce_many()."""
    L = _lib.lib()
    n = xs[0].shape[0]
    ws = _lib.scratch(max(L.linr_linear_wgrad_wide_workspace_bytes(n, cin, cout), 4), xs[0].device)
    deferred = defer is not None and n > 0 and not accumulate
    check(L.linr_linear_wgrad_wide(_ptr_array(xs), cin, 1 if in_blocked else 0, _ptr_array(gouts), cout, 1 if g_blocked else 0, n,
                                   None if deferred else gw.data_ptr(), ws_ci, ws_co, _ptr(gb), LINR_ACCUM if accumulate else 0,
                                   ws.data_ptr(), ws.numel(), _stream()), 'linr_linear_wgrad_wide')
    if deferred:
        defer.append((_lib.LinrWideReduce(1, int(L.linr_linear_wgrad_wide_blocks(n)), cin, cout, ws_ci, ws_co, ws.data_ptr(), gw.data_ptr(),
                                          _ptr(gb)), ws, gw, gb))


def head_wide_fwd(cs, w1, b1, w2, b2, target, p, bits=None, partial=None):
    """linr_head_wide_fwd: p = sigmoid(w2 . relu(W1 c + b1) + b2) of a wide head on the blocks cs; bits (float64[1]) += the stage's bits
    against the occupancy column `target` (a strided view) when given; partial (float64 tensor, >= linr_head_wide_workspace_bytes / 8
    elements) instead: the stage's per-block partials stay there for bits_finish()."""
    L = _lib.lib()
    n, C = cs[0].shape[0], 8 * len(cs)
    ws = partial
    if bits is not None and ws is None:
        ws = _lib.scratch(max(L.linr_head_wide_workspace_bytes(n), 8), cs[0].device)
    check(L.linr_head_wide_fwd(_ptr_array(cs), C, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), _ptr(target),
                               1 if target is None else target.stride(0), n, p.data_ptr(), _ptr(bits) if partial is None else None,
                               _ptr(ws), 0 if ws is None else ws.numel() * ws.element_size(), _stream()), 'linr_head_wide_fwd')
    return p


def bits_finish(partial, count, bits):
    """linr_bits_finish: bits (float64[1]) += the sum of the first `count` partials / ln 2."""
    check(_lib.lib().linr_bits_finish(partial.data_ptr(), count, bits.data_ptr(), _stream()), 'linr_bits_finish')


def head_wide_bwd(cs, ps, targets, w1s, b1s, w2s, gscale, gcs, grads):
    """linr_head_wide_bwd: the backward of len(ps) wide heads in one grouped launch.  cs / gcs: per stage the list of C / 8 blocks;
    grads: the contiguous [stages][W1 | b1 | w2 | b2] gradient region (a view of the flat gradient)."""
    import ctypes
    L = _lib.lib()
    ns, nb = len(ps), len(cs[0])
    n, C = cs[0][0].shape[0], 8 * nb

    def arr(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    slab = _lib.scratch(L.linr_head_wide_bwd_slab_bytes(C, ns), cs[0][0].device)
    assert grads.is_contiguous() and grads.numel() == ns * (24 * C + 49)
    check(L.linr_head_wide_bwd(arr([b for c in cs for b in c]), arr(ps), arr(targets), targets[0].stride(0), arr(w1s), arr(b1s), arr(w2s),
                               C, ns, float(gscale), arr([b for g in gcs for b in g]), n, slab.data_ptr(), slab.numel(), grads.data_ptr(),
                               _stream()), 'linr_head_wide_bwd')


def kmap_tile8t(nbr, n=None):
    """linr_kmap_tile8t: the tiled copy of the kernel map in the lane order of the transposing weight-gradient kernel."""
    L = _lib.lib()
    n = nbr.shape[1] if n is None else n
    out = torch.empty(max(4, (L.linr_kmap_tile8t_bytes(n) + 3) // 4), dtype=torch.int32, device=nbr.device)
    check(L.linr_kmap_tile8t(nbr.data_ptr(), nbr.stride(0), n, out.data_ptr(), out.numel() * 4, _stream()), 'linr_kmap_tile8t')
    return out


def spconv_wgrad_cmap(x, gout, nbr, n, cin, cout, slab=None, reduce=True, tile8t=None):
    """The stand-alone backward-weight kernel (MFMA): with tile8t (linr_kmap_tile8t) the transposing kernel with coalesced gathers,
    without it the direct-gather kernel reading nbr.  x: view buf[1:] of a [n+1, 8] buffer whose row 0 is zero.
    Returns (gW [27,cin,cout], gb [cout]) summed over the per-block partials, or the raw slab with reduce=False."""
    L = _lib.lib()
    nb = int(L.linr_spconv_wgrad_cmap_blocks())
    elems = (27 * cin + 1) * cout
    if slab is None:
        slab = _lib.scratch(nb * elems * 4, x.device).view(torch.float32).view(nb, elems)
    check(L.linr_spconv_wgrad_cmap(x.data_ptr(), x.stride(0), gout.data_ptr(), gout.stride(0), nbr.data_ptr(), _ptr(tile8t),
                                   nbr.stride(0), n, cin, cout, slab.data_ptr(), _stream()),
          'linr_spconv_wgrad_cmap')
    if not reduce:
        return slab
    gw = torch.empty((27, cin, cout), dtype=torch.float32, device=x.device)
    gb = torch.empty((cout,), dtype=torch.float32, device=x.device)
    check(L.linr_slab_reduce(slab.data_ptr(), nb, elems, 27 * cin * cout, gw.data_ptr(), gb.data_ptr(), 0, _stream()), 'linr_slab_reduce')
    return gw, gb


def spconv_bwd_fused(gout, x, lo, mask, n, kernel, nblocks=256, reduce=True):
    """linr_spconv_bwd_fused: backward-data and weight gradient of a conv 8->8 from one gather of the output gradient.
    gout: view buf[1:] of a [n+1, 8] buffer whose row 0 is zero; x [n, 8] the convolution's input.
    Returns (gin [n, 8], gW [27, 8, 8], gb [8]), or (gin, slab [nblocks, 1736]) with reduce=False."""
    gin = torch.empty((n, 8), dtype=torch.float32, device=gout.device)
    slab = torch.full((nblocks, 1736), float('nan'), dtype=torch.float32, device=gout.device)
    check(_lib.lib().linr_spconv_bwd_fused(gout.data_ptr(), x.data_ptr(), lo.data_ptr(), mask.data_ptr(), lo.stride(0), n,
                                           kernel.data_ptr(), gin.data_ptr(), slab.data_ptr(), nblocks, _stream()),
          'linr_spconv_bwd_fused')
    if not reduce:
        return gin, slab
    tot = slab.double().sum(dim=0).float()
    return gin, tot[:1728].view(27, 8, 8), tot[1728:]


def occ_wgrad7(occ, gouts, lo, mask, n, nblocks=256):
    """linr_occ_wgrad7: weight gradients of the first convolutions of the 7 outter blocks from one gather of the occupancy rows.
    occ: view buf[1:] of a [n+1, 8] buffer whose row 0 is zero; gouts: 7 tensors [n, 8].
    Returns ([gW_b [27, b, 8] for b = 1..7], [gb_b [8]]) summed over the slab rows the kernel wrote."""
    import ctypes
    assert len(gouts) == 7
    slab = torch.full((nblocks, 6104), float('nan'), dtype=torch.float32, device=occ.device)
    ptrs = (ctypes.c_void_p * 7)(*[g.data_ptr() for g in gouts])
    rows = ctypes.c_int32(0)
    check(_lib.lib().linr_occ_wgrad7(occ.data_ptr(), ptrs, lo.data_ptr(), mask.data_ptr(), lo.stride(0), n, slab.data_ptr(), nblocks,
                                     ctypes.byref(rows), _stream()), 'linr_occ_wgrad7')
    tot = slab[:rows.value].double().sum(dim=0).float()
    gw, gb, cur = [], [], 0
    for b in range(1, 8):
        gw.append(tot[cur:cur + 27 * b * 8].view(27, b, 8)); cur += 27 * b * 8
        gb.append(tot[cur:cur + 8]); cur += 8
    return gw, gb


def _aligned_ws(nbytes, device):
    ws = _lib.scratch(nbytes + 256, device)
    base = (ws.data_ptr() + 255) & ~255
    return ws, base, ws.numel() - (base - ws.data_ptr())


def coords_minmax(coords):
    """Per-axis (min x, y, z, max x, y, z) of an int32 [n,3] GPU coordinate list as a device int32 [6] tensor (two launches, no host read)."""
    _dev(coords, torch.int32, 'coords')
    out = torch.empty(6, dtype=torch.int32, device=coords.device)
    check(_lib.lib().linr_coords_minmax(coords.data_ptr(), coords.shape[0], out.data_ptr(), _stream()), 'linr_coords_minmax')
    return out


def coords_sort_unique(coords, shift=0, coord_bits=20, origin=None):
    """Sorted (x-major) unique rows of ((coords - origin) >> shift): int32 [n,3] on the GPU in, int32 [m,3] out; shifted coordinates
    in [0, 2^coord_bits); origin: int32 [3] GPU tensor or None.  One library call (linr_coords_sort_unique) + one host read of the
    count - torch.unique(dim=0) of the reference's dataset code (custom_dataset.py:271-282, shift 0)."""
    _dev(coords, torch.int32, 'coords')
    L = _lib.lib()
    n = coords.shape[0]
    out = torch.empty((n, 3), dtype=torch.int32, device=coords.device)
    if n == 0:
        return out
    count = torch.empty(1, dtype=torch.int64, device=coords.device)
    ws, base, nbytes = _aligned_ws(L.linr_sort_unique_workspace_bytes(n), coords.device)
    check(L.linr_coords_sort_unique(coords.data_ptr(), n, None if origin is None else origin.data_ptr(), int(shift), int(coord_bits),
                                    out.data_ptr(), count.data_ptr(), base, nbytes, _stream()), 'linr_coords_sort_unique')
    m = int(count)
    return out if m == n else out[:m].clone()


def octree_level(child, coord_bits=20):
    """One octree level (octree_level.forward, models/module_utils.py:86-110) in one library call: child int32 [m,3] sorted x-major
    and unique -> (parent int32 [n,3] = sorted unique child >> 1, occ float32 [n,8])."""
    _dev(child, torch.int32, 'child')
    L = _lib.lib()
    m = child.shape[0]
    parent = torch.empty((m, 3), dtype=torch.int32, device=child.device)
    occ = torch.empty((m, 8), dtype=torch.float32, device=child.device)
    if m == 0:
        return parent, occ
    count = torch.empty(1, dtype=torch.int64, device=child.device)
    ws, base, nbytes = _aligned_ws(L.linr_sort_unique_workspace_bytes(m), child.device)
    check(L.linr_octree_level(child.data_ptr(), m, int(coord_bits), parent.data_ptr(), occ.data_ptr(), count.data_ptr(), base, nbytes,
                              _stream()), 'linr_octree_level')
    n = int(count)
    return parent[:n].clone(), occ[:n].clone()          # exact-size copies: the m-row buffers go back to the allocator


def octree_levels(child, coord_bits, max_levels, count_dev=None):
    """All octree levels below a sorted unique child list in ONE library call (linr_octree_levels: bitmap of the parents' compact keys,
    no sort, counts chained on the device) and ONE host read.  child: int32 [m,3] on the GPU, coordinates in [0, 2^coord_bits),
    coord_bits <= 11; count_dev: None or the device int64 live row count (<= m).  Returns (parents int32 [total,3], occ float32
    [total,8], counts list): level l's rows are parents[sum(counts[:l]) : sum(counts[:l + 1])] - exact-size buffers shared by the
    levels - or None when the library has no such levels (coord_bits out of range)."""
    _dev(child, torch.int32, 'child')
    L = _lib.lib()
    m = child.shape[0]
    nlev = L.linr_octree_levels_count(int(coord_bits), int(max_levels))
    if nlev < 1 or m == 0:
        return None
    cap = L.linr_octree_levels_rows(m, int(coord_bits), int(max_levels))
    parents = torch.empty((cap, 3), dtype=torch.int32, device=child.device)
    occ = torch.empty((cap, 8), dtype=torch.float32, device=child.device)
    counts = torch.empty(nlev, dtype=torch.int64, device=child.device)
    ws, base, nbytes = _aligned_ws(L.linr_octree_levels_workspace_bytes(m, int(coord_bits), int(max_levels)), child.device)
    check(L.linr_octree_levels(child.data_ptr(), m, None if count_dev is None else count_dev.data_ptr(), int(coord_bits), int(max_levels),
                               parents.data_ptr(), occ.data_ptr(), counts.data_ptr(), base, nbytes, _stream()), 'linr_octree_levels')
    ch = counts.tolist()                                  # the one host read
    total = int(sum(ch))
    return parents[:total].clone(), occ[:total].clone(), ch


def octree_occupancy(child, parent):
    """occ float32 [N,8] of the parents (sorted unique floor(child/2)) of a sorted unique child list (int32 [M,3])."""
    _dev(child, torch.int32, 'child')
    _dev(parent, torch.int32, 'parent')
    L = _lib.lib()
    m, n = child.shape[0], parent.shape[0]
    occ = torch.empty((n, 8), dtype=torch.float32, device=child.device)
    ws = _lib.scratch(max(L.linr_kmap_workspace_bytes(m), 8), child.device)
    check(L.linr_octree_occupancy(child.data_ptr(), m, parent.data_ptr(), n, occ.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          'linr_octree_occupancy')
    return occ
