"""LINR_PCGC_Model: drop-in mirror of models/model_core.py on the MI355X engine.

Same constructor dict, same state_dict (189 tensors, reference order), same call surface
(``forward -> bits``, ``codec``, ``encode``, ``decode``).  All network arithmetic is in HIP kernels
(csrc/*.hip) behind include/linr_hip.h; this class owns the parameters (as views into ONE flat buffer, the layout
the kernels index), caches one kernel map + arena per coordinate set, and feeds the C++ range coder.

Two ways in:
  * reference-style: ``bits = model(putin_args)`` per scale, ``loss.backward()``, any torch optimiser
    (main.py:457-475,305-321) - autograd is bridged by ``_NetBits``;
  * fast path used by ``overfit.py`` / ``bench.py``: ``Frame`` with all scales batched + ``train_step`` (forward,
    backward and the fused Adam update without touching autograd).
"""
import time

import numpy as np
import torch
from torch import nn

from . import _lib, engine, ops
from .function_utils import pack_bitstream, unpack_bitstream
from .module_utils import BinaryArithmeticCoding, PointwiseMLP
from .upsample import CNP


class _NetBits(torch.autograd.Function):
    """bits(frame; params) with the hand-written backward of csrc/net.hip.  Parameter gradients are accumulated straight
    into the model's flat gradient buffer (the 189 ``.grad`` tensors are views of it)."""

    @staticmethod
    def forward(ctx, model, frame, *params):
        bits = torch.zeros(1, dtype=torch.float64, device=frame.device)
        engine.net_forward(frame, model._flat, 0, 8, None, bits)
        ctx.model, ctx.frame = model, frame
        return bits[0].to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        model, frame = ctx.model, ctx.frame
        tmp = torch.zeros_like(model._flat)
        engine.net_backward(frame, model._flat, tmp, 1.0)
        model._ensure_grad_views()
        model._flat_grad.addcmul_(tmp, gout.to(torch.float32).reshape(1).expand_as(tmp))
        return (None, None) + (None,) * len(model._plist)


class _WideBits(torch.autograd.Function):
    """The same for hidden_channel_conv = 16 / 32 (wide_net.py: forward with the activations kept, hand-written backward)."""

    @staticmethod
    def forward(ctx, model, frame, *params):
        bits = torch.zeros(1, dtype=torch.float64, device=frame.device)
        with torch.no_grad():
            ctx.tape = model._wide.forward(frame, 0, 8, None, bits, keep=True)
        ctx.model, ctx.frame = model, frame
        return bits[0].to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        model, frame = ctx.model, ctx.frame
        model._ensure_grad_views()
        keep = model._flat_grad.clone()                      # the wide backward WRITES the gradients: accumulate around it
        with torch.no_grad():
            model._flat_grad.zero_()
            if ctx.tape is not None:
                model._wide.backward(frame, ctx.tape, 1.0)
            model._flat_grad.mul_(gout.to(torch.float32).reshape(1)).add_(keep)
        return (None, None) + (None,) * len(model._plist)


class LINR_PCGC_Model(nn.Module):
    """models/model_core.py:19-37.  inargs: scale_num, in_channel (=7), hidden_channel_conv (=8), block_layers (1..4),
    outstage (=8), instage (=1).  outstage / instage / in_channel are hard-coded by the reference's drivers
    (main.py:97,218); block_layers is its live --block_layers flag (main.py:521: the Inception layers of block_in's
    ResNetBlock, with the extra skip of models/resnet.py:160-161 when > 1).  hidden_channel_conv (main.py:520): 8 runs on the
    tuned kernels (every one is specialised for 8-wide feature rows), 16 and 32 on the channel-blocked executor of wide_net.py."""

    def __init__(self, inargs):
        super().__init__()
        self.scale_num = int(inargs['scale_num'])
        if not 1 <= self.scale_num <= 16:
            raise ValueError('scale_num must be in 1..16 (the executor batches at most 16 scales per frame; the reference runs 6 to 8), '
                             'got %d' % self.scale_num)
        in_channel = int(inargs['in_channel'])
        hidden = int(inargs['hidden_channel_conv'])
        block_layers = int(inargs['block_layers'])
        if in_channel != 7 or inargs['outstage'] != 8 or inargs['instage'] != 1:
            raise ValueError('the gfx950 engine is specialised for in_channel=7, outstage=8, instage=1 (what main.py:97,218 '
                             'hard-code); got %r' % (inargs,))
        if hidden not in (8, 16, 32):
            raise ValueError('hidden_channel_conv=%d is not supported: 8 (the reference default, main.py:520) runs on the tuned '
                             'gfx950 kernels, 16 and 32 on the channel-blocked executor (wide_net.py); a checkpoint of another '
                             'width cannot be loaded' % hidden)
        self.hidden = hidden
        if not 1 <= block_layers <= 4:
            raise ValueError('block_layers must be in 1..4 (main.py:521 default 1), got %d' % block_layers)
        self.block_layers = block_layers
        self.scale_emb = nn.Embedding(self.scale_num, 8)
        self.scale_mlp = nn.ModuleList([PointwiseMLP([8 + in_channel, 16, 8]) for _ in range(self.scale_num)])
        self.upsampler = CNP(in_channels=8, channels=hidden, block_layers=block_layers, outstage=8, instage=1)
        self.sigmoid = nn.Sigmoid()
        self._flat = None
        self._flat_grad = None
        self._plist = []
        self._frame_cache = {}
        # inference numerics of frame_probs / codec / encode / decode: 'f32' (default) or 'bf16' = BASELINE config[4]'s bf16
        # features + uint8 weight codes (needs set_quantised, which the model codec calls)
        self.inference_precision = 'f32'
        # arithmetic of train_step: 'f32' (default, the headline) or 'bf16' = bf16 feature / gradient rows with fp32 master
        # parameters and accumulation (linr_net_train_step_bf16; hidden_channel_conv 8, block_layers 1)
        self.train_precision = 'f32'
        self._qcodes = None
        self._qrange = None
        self._wide = None
        if hidden != 8:
            from .wide_net import WideNet
            self._wide = WideNet(self, hidden)
        self._flatten()

    # ---- flat parameter buffer ------------------------------------------------------------------------------------
    def _flatten(self):
        """Re-home all parameters as views of one contiguous buffer in parameters() order (the kernels' layout)."""
        plist = list(self.parameters())
        total = sum(p.numel() for p in plist)
        if plist and plist[0].is_cuda and self.hidden == 8 and total != _lib.lib().linr_param_count(self.scale_num, self.block_layers):
            raise _lib.LinrError('parameter layout mismatch with liblinr_hip.so')
        flat = torch.empty(total, dtype=torch.float32, device=plist[0].device)
        off = 0
        with torch.no_grad():
            for p in plist:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = None
                off += n
        self._flat, self._plist, self._flat_grad = flat, plist, None
        self._frame_cache = {}
        if getattr(self, '_wide', None) is not None:
            self._wide._built = False                 # the parameter tensors were re-homed

    def _apply(self, fn, *args, **kwargs):
        # .cuda() / .to(device) / .cpu(): ONE conversion of the flat buffer and 189 new views instead of nn.Module's per-tensor
        # conversion followed by a second flattening (8.5 -> 3 ms per model; the drivers build three models per GOP).  Anything that
        # changes the dtype, and modules with buffers, take the general road.
        flat = None
        if self._flat is not None and not any(True for _ in self.buffers()):
            with torch.no_grad():
                flat = fn(self._flat)
            if flat.dtype != torch.float32 or flat.numel() != self._flat.numel() or not flat.is_contiguous():
                flat = None
        if flat is None:
            out = super()._apply(fn, *args, **kwargs)
            self._flatten()
            return out
        if flat.is_cuda and self.hidden == 8 and flat.numel() != _lib.lib().linr_param_count(self.scale_num, self.block_layers):
            raise _lib.LinrError('parameter layout mismatch with liblinr_hip.so')
        off = 0
        with torch.no_grad():
            for q in self._plist:
                n = q.numel()
                q.data = flat[off:off + n].view(q.shape)
                q.grad = None
                off += n
        self._flat, self._flat_grad, self._frame_cache = flat, None, {}
        if self._qcodes is not None:
            self._qcodes = self._qcodes.to(flat.device)
        if getattr(self, '_wide', None) is not None:
            self._wide._built = False                 # the parameter tensors were re-homed
        return self

    def _ensure_grad_views(self):
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
        off = 0
        for p in self._plist:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self._flat_grad.data_ptr() + 4 * off:
                if p.grad is None:
                    self._flat_grad[off:off + n].zero_()        # zero_grad(set_to_none=True) semantics
                else:
                    self._flat_grad[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = self._flat_grad[off:off + n].view(p.shape)
            off += n

    def set_quantised(self, codes, min_param, max_param):
        """The model as its uint8 codes of quant_uniform2 (model_compression/model_size_est.py:72-91): what model.bin holds.
        Called by Model_Estimate.compress_model / decompress_model next to the de-quantised fp32 fill."""
        if codes.numel() != self._flat.numel():
            raise ValueError('expected %d codes, got %d' % (self._flat.numel(), codes.numel()))
        self._qcodes = codes.to(device=self._flat.device, dtype=torch.uint8).contiguous()
        self._qrange = (float(min_param), float(max_param))

    def _precision(self, precision):
        precision = self.inference_precision if precision is None else precision
        if precision not in ('f32', 'bf16'):
            raise ValueError("precision must be 'f32' or 'bf16'")
        if precision == 'bf16' and self._qcodes is None:
            raise _lib.LinrError('the bf16 path runs from the 8-bit weight codes: quantise the model first, at bitdepth 8 '
                                 '(Model_Estimate.compress_model(model, 8, derive_new_model=True) / decompress_model)')
        return precision

    def flat_parameters(self):
        """The contiguous float32 buffer holding every parameter in parameters() order (what the model codec codes)."""
        return self._flat

    # ---- frames ----------------------------------------------------------------------------------------------------
    def make_frame(self, scales, validate=True, with_arena=True):
        """Batched multi-scale frame for the fast path.  scales: list of per-scale input dicts (see engine.Frame)."""
        return engine.Frame(scales, self.scale_num, self._flat.device, validate, with_arena and self._wide is None, self.block_layers)

    def _scale_frame(self, d, need_occ=True):
        """Kernel map + arena for one scale's inputs.  Encoder-side inputs are cached by tensor identity; the cache
        entry pins the tensors so a recycled allocation can never alias a stale kernel map.  Decoder-side frames
        (fresh coordinates every call) are not cached."""
        coord = d['coord']
        s = {'coord': coord, 'offset_tensor': d.get('offset_tensor'), 'scale_idx': d['scale_idx']}
        if not need_occ:
            return self.make_frame([s])
        occ_lst, off = d.get('occ_lst'), d.get('offset_tensor')
        if occ_lst is None:                                  # the GOP drivers' lean dicts carry the matrix only (module_utils.prepare_frame)
            occ_lst = list(d['occ'].split(1, dim=1))
        occ0 = occ_lst[0]
        # identity AND shape of every input: slices of the cached tensors share their data pointers
        key = (coord.data_ptr(), tuple(coord.shape), int(d['scale_idx']), None if off is None else (off.data_ptr(), tuple(off.shape)),
               occ0.data_ptr(), len(occ_lst), tuple(tuple(o.shape) for o in occ_lst))
        hit = self._frame_cache.get(key)
        if hit is None:
            if len(self._frame_cache) >= 1024:
                self._frame_cache.pop(next(iter(self._frame_cache)))
            s['occ_lst'] = occ_lst
            hit = (self.make_frame([s]), coord, off, list(occ_lst))
            self._frame_cache[key] = hit
        return hit[0]

    # ---- reference call surface --------------------------------------------------------------------------------------
    def forward(self, inargs):
        """models/model_core.py:72-81: bits of one scale (0-dim float32, differentiable w.r.t. the parameters)."""
        if int(inargs['coord'].shape[0]) == 0:          # a scale without voxels costs nothing (and has nothing to differentiate)
            return self._flat.sum() * 0.0
        frame = self._scale_frame(inargs)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._plist):
            return (_WideBits if self._wide is not None else _NetBits).apply(self, frame, *self._plist)
        bits = torch.zeros(1, dtype=torch.float64, device=frame.device)
        self._stage_forward(frame, 0, 8, None, bits, 'f32')
        return bits[0].to(torch.float32)

    def _stage_forward(self, frame, k0, k1, probs, bits, precision):
        if self._wide is not None:
            if precision != 'f32':
                raise _lib.LinrError('the bf16 / uint8-weight executor exists for hidden_channel_conv=8 only')
            with torch.no_grad():
                self._wide.forward(frame, k0, k1, probs, bits)
        elif precision == 'bf16':
            engine.net_forward_bf16(frame, self._qcodes, self._qrange[0], self._qrange[1], k0, k1, probs, bits)
        else:
            engine.net_forward(frame, self._flat, k0, k1, probs, bits)

    @torch.no_grad()
    def frame_probs(self, frame, precision=None):
        """One inference forward over a (possibly multi-scale) frame: probs float32 [8, rows], bits float64[1]."""
        probs = torch.empty((8, frame.rows), dtype=torch.float32, device=frame.device)
        bits = torch.zeros(1, dtype=torch.float64, device=frame.device)
        self._stage_forward(frame, 0, 8, probs, bits, self._precision(precision))
        return probs, bits

    @torch.no_grad()
    def codec(self, inargs):
        """models/model_core.py:169-227: one forward, the 8 stages as ONE arithmetic-coded stream, timed enc/dec."""
        st1 = time.time()
        frame = self._scale_frame(inargs)
        probs, bits_t = self.frame_probs(frame)
        p_host = probs.reshape(-1).cpu()
        sym = frame.occ.t().contiguous().reshape(-1).to(torch.int16).cpu()
        bac = BinaryArithmeticCoding()
        st2 = time.time()
        enc_bytes = bac.encode(p_host, sym)
        st3 = time.time()
        recon = bac.decode(p_host, enc_bytes)
        st4 = time.time()
        assert bool((recon == sym).all())
        return {'bits': len(enc_bytes) * 8, 'enc_bytes': enc_bytes, 'enc_time': st3 - st1,
                'dec_time': (st2 - st1) + (st4 - st3), 'bits_t': bits_t[0].to(torch.float32)}

    @torch.no_grad()
    def encode(self, in_args, DBG=False):
        """models/model_core.py:235-266 + CNP.encode (models/upsample.py:219-246): 8 per-stage streams, packed."""
        frame = self._scale_frame(in_args)
        probs, _ = self.frame_probs(frame)
        p_host = probs.cpu().numpy()
        occ_host = frame.occ.t().contiguous().cpu().numpy().astype(np.uint8)
        streams = encode_streams([p_host[k] for k in range(8)], [occ_host[k] for k in range(8)])
        if DBG:
            bac = BinaryArithmeticCoding()
            for k in range(8):
                assert (bac.decode(p_host[k], streams[k]).numpy() == occ_host[k]).all()
        enc_bytes = pack_bitstream(streams)
        return {'enc_bytes': enc_bytes, 'bits': len(enc_bytes) * 8, 'x_low': None}

    @torch.no_grad()
    def decode(self, inagrs):
        """models/model_core.py:268-286 + CNP.decode (models/upsample.py:249-295): stage-serial, the SAME launches as
        the encoder's forward, so probabilities are bitwise identical.  Returns 8 x [N,1] float32 occupancy."""
        streams = unpack_bitstream(inagrs['enc_bytes'])
        s = {'coord': inagrs['coord'], 'offset_tensor': inagrs.get('offset_tensor'), 'scale_idx': inagrs['scale_idx']}
        # decoder-side frames are never cached (fresh coordinates every call); the bf16 path brings its own small arena
        frame = self.make_frame([s], with_arena=self._precision(None) == 'f32')
        return self.decode_frame(frame, [streams])

    @torch.no_grad()
    def decode_scale(self, coord, scale_idx, enc_bytes, child_bits):
        """One scale of decoder.decode_one_frame (decoder.py:153-176) as ONE C call that does not hold the GIL
        (linr_decode_scale): kernel map of `coord` (int32 [n,3] on the GPU, sorted x-major), the 8 decode stages against the
        packed stream `enc_bytes`, octree_level.upper_layer.  Returns the next finer level's coordinates (int32 [m,3])."""
        import ctypes
        L = _lib.lib()
        n = int(coord.shape[0])
        if n == 0:
            return coord.new_zeros((0, 3))
        coord = coord.contiguous()
        streams = unpack_bitstream(enc_bytes)
        bufs = [np.frombuffer(b, dtype=np.uint8) for b in streams]
        ptrs = (ctypes.c_void_p * 8)(*[b.ctypes.data if b.size else None for b in bufs])
        lens = (ctypes.c_int64 * 8)(*[int(b.size) for b in bufs])
        bf16 = self._precision(None) == 'bf16'
        need = L.linr_decode_scale_ws_bytes(n, self.block_layers, 1 if bf16 else 0)
        ws = _lib.scratch(need + 256, coord.device)
        base = (ws.data_ptr() + 255) & ~255
        child = torch.empty((8 * n, 3), dtype=torch.int32, device=coord.device)
        p_host, s_host = self._host_buffers(n)
        m = ctypes.c_int64(0)
        codes, lo, hi, params = (self._qcodes.data_ptr(), float(self._qrange[0]), float(self._qrange[1]), None) if bf16 else \
            (None, 0.0, 0.0, self._flat.data_ptr())
        _lib.check(L.linr_decode_scale(coord.data_ptr(), n, int(scale_idx), self.scale_num, self.block_layers, int(child_bits), params,
                                       codes, lo, hi, ptrs, lens, base, need, p_host.data_ptr(), s_host.data_ptr(),
                                       child.data_ptr(), 8 * n, ctypes.byref(m), torch.cuda.current_stream().cuda_stream),
                   'linr_decode_scale')
        return child[:m.value]

    def _host_buffers(self, rows):
        """Pinned staging buffers of the staged decoder (probabilities down, decoded symbols up), grown on demand."""
        import threading
        tls = self.__dict__.setdefault('_dec_tls', threading.local())          # one pair per decoding thread
        buf = getattr(tls, 'buf', None)
        if buf is None or buf[0].numel() < rows:
            cap = max(rows, 1 << 16)
            buf = (torch.empty(cap, dtype=torch.float32, pin_memory=True), torch.empty(cap, dtype=torch.uint8, pin_memory=True))
            tls.buf = buf
        return buf

    @torch.no_grad()
    def decode_frame(self, frame, streams_per_scale, precision=None):
        """Staged decode of every scale of `frame` at once: stage k of all scales is one launch set.  Host side per
        stage: one D2H of the probabilities into pinned memory, linr_ac_decode_binary per scale straight on those
        buffers, one H2D of the decoded byte column (no torch CPU ops: they fan out over every host core)."""
        frame.occ.zero_()
        frame.invalidate_occ()
        rows = frame.rows
        probs = torch.empty((8, rows), dtype=torch.float32, device=frame.device)
        s_dev = torch.empty(max(rows, 1), dtype=torch.uint8, device=frame.device)
        p_host, s_host = self._host_buffers(rows)
        precision = self._precision(precision)
        if self._wide is not None:          # stage loop in Python: stage forward, D2H, range decoder, H2D
            L = _lib.lib()
            import ctypes
            for k in range(8):
                self._stage_forward(frame, k, k + 1, probs, None, precision)
                p_host[:rows].copy_(probs[k])
                for i in range(frame.n_scales):
                    r0, r1 = int(frame.row_off[i]), int(frame.row_off[i + 1])
                    if r1 == r0:
                        continue
                    buf = np.frombuffer(streams_per_scale[i][k], dtype=np.uint8)
                    _lib.check(L.linr_ac_decode_binary(ctypes.c_void_p(p_host.data_ptr() + 4 * r0), r1 - r0,
                                                       ctypes.c_void_p(buf.ctypes.data if buf.size else None), int(buf.size),
                                                       ctypes.c_void_p(s_host.data_ptr() + r0)), 'linr_ac_decode_binary')
                frame.occ[:, k].copy_(s_host[:rows].to(frame.device, torch.float32))
            return [frame.occ[:, k:k + 1].clone() for k in range(8)]
        # the whole stage loop is one C call (csrc/net.hip: linr_net_decode_stages): no Python between the stages, no GIL held
        if precision == 'bf16':
            engine.net_decode_stages(frame, None, streams_per_scale, probs, p_host, s_host, s_dev, self._qcodes, self._qrange)
        else:
            engine.net_decode_stages(frame, self._flat, streams_per_scale, probs, p_host, s_host, s_dev)
        return [frame.occ[:, k:k + 1].clone() for k in range(8)]


def encode_streams(probs, symbols, n_threads=8):
    """Codes independent binary streams on a host thread pool (linr_ac_encode_binary_batch)."""
    import ctypes
    n = len(probs)
    probs = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1) for p in probs]
    symbols = [np.ascontiguousarray(s, dtype=np.uint8).reshape(-1) for s in symbols]
    outs = [np.empty(2 * p.size + 64, dtype=np.uint8) for p in probs]
    arr_p = (ctypes.c_void_p * n)(*[p.ctypes.data for p in probs])
    arr_s = (ctypes.c_void_p * n)(*[s.ctypes.data for s in symbols])
    arr_o = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    arr_n = (ctypes.c_int64 * n)(*[p.size for p in probs])
    arr_c = (ctypes.c_int64 * n)(*[o.size for o in outs])
    arr_l = (ctypes.c_int64 * n)()
    _lib.check(_lib.lib().linr_ac_encode_binary_batch(arr_p, arr_s, arr_n, n, arr_o, arr_c, arr_l, n_threads),
               'linr_ac_encode_binary_batch')
    return [outs[i][:arr_l[i]].tobytes() for i in range(n)]


class FlatAdam:
    """torch.optim.Adam(lr, betas, eps, L2 weight_decay) + StepLR(step_size, gamma) of main.py:231-252,319-321 over the
    model's flat parameter buffer: one fused launch per step (linr_adam_step).  state_dict()/load_state_dict() use
    torch.optim.Adam's format so reference checkpoints (``optimizer_state_dict``) round-trip."""

    def __init__(self, model, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, step_size=32, gamma=0.992):
        self.model = model
        self.lr, self.initial_lr = float(lr), float(lr)
        self.betas, self.eps, self.weight_decay = tuple(betas), float(eps), float(weight_decay)
        self.step_size, self.gamma = int(step_size), float(gamma)
        flat = model.flat_parameters()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.grad = torch.zeros_like(flat)
        self.t = 0                      # optimiser steps taken
        self.sched_steps = 0            # scheduler.step() calls (StepLR epoch counter)
        # torch.optim.Adam keeps one step counter per parameter and skips a parameter whose .grad is None.  The reference pins
        # torch 1.13.1 (enviroment.yaml:30), whose optimizer.zero_grad() (main.py:320) leaves ZERO tensors behind, not None: the
        # context MLP of a scale is therefore skipped only until a frame containing that scale has given it its first gradient
        # (custom_dataset.py:325 drops the coarsest scales of small frames); from then on it is updated on every step, with a
        # zero gradient on frames that lack the scale (weight decay and moment decay still act, its counter advances).
        # t_scale[s] = Adam updates applied to scale s so far (0: never had a gradient).
        self.t_scale = np.zeros(model.scale_num, dtype=np.int64)
        self._segments = None

    def zero_grad(self):
        self.grad.zero_()

    def reset(self):
        """Back to the state of a freshly constructed optimiser, in place (no allocation, no host round trip)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.grad.zero_()
        self.lr = self.initial_lr
        self.t = 0
        self.sched_steps = 0
        self.t_scale[:] = 0

    def advance(self, frame=None):
        """Step counters of the next update: (t, t_scale) after an optimiser step on `frame` (None: a dense gradient, every scale
        present).  Scales of the frame start their counter; started scales advance on every step (see __init__)."""
        t_scale = self.t_scale.copy()
        if frame is None:
            t_scale += 1
        else:
            present = np.zeros(len(t_scale), dtype=bool)
            for i in range(frame.n_scales):
                if frame.row_off[i + 1] > frame.row_off[i]:
                    present[frame.scale_idx[i]] = True
            t_scale[present | (t_scale > 0)] += 1
        return self.t + 1, t_scale

    def _param_segments(self):
        """[(begin, end, scale or -1)] over the flat buffer: maximal runs of parameters with the same owner."""
        if self._segments is None:
            segs, off = [], 0
            for own, p in zip(self._scale_of_param(), self.model._plist):
                n = p.numel()
                if segs and segs[-1][2] == own:
                    segs[-1][1] = off + n
                else:
                    segs.append([off, off + n, own])
                off += n
            self._segments = [tuple(x) for x in segs]
        return self._segments

    def step(self, frame=None):
        """torch.optim.Adam.step() on the dense gradient buffer self.grad: the same update, segment by segment, that the fused
        train_step applies in one launch (a scale's context MLP is skipped until its first gradient; `frame` says which scales
        the gradient came from, None = all)."""
        t, t_scale = self.advance(frame)
        flat = self.model.flat_parameters()
        runs = []                                   # adjacent segments with the same step count are one launch (usually: all of them)
        for b, e, own in self._param_segments():
            ts = t if own < 0 else int(t_scale[own])
            if ts < 1:
                continue
            if runs and runs[-1][1] == b and runs[-1][2] == ts:
                runs[-1][1] = e
            else:
                runs.append([b, e, ts])
        for b, e, ts in runs:
            ops.adam_step(flat[b:e], self.grad[b:e], self.exp_avg[b:e], self.exp_avg_sq[b:e], ts, self.lr,
                          self.betas[0], self.betas[1], self.eps, self.weight_decay)
        self.t, self.t_scale = t, t_scale

    def scheduler_step(self):
        """StepLR.step(): multiply lr by gamma every `step_size` calls (chainable form, so a clamp persists)."""
        self.sched_steps += 1
        if self.sched_steps % self.step_size == 0:
            self.lr *= self.gamma

    def clamp_lr(self, min_lr):
        """main.py:433-437."""
        if self.lr < min_lr:
            self.lr = min_lr

    def _scale_of_param(self):
        """parameter index -> scale of its scale_mlp, or -1 (names: scale_mlp.<s>.<layer>.<weight|bias>)."""
        out = []
        for name, _ in self.model.named_parameters():
            out.append(int(name.split('.')[1]) if name.startswith('scale_mlp.') else -1)
        return out

    def state_dict(self):
        state, off = {}, 0
        owner = self._scale_of_param()
        for i, p in enumerate(self.model._plist):
            n = p.numel()
            t = self.t if owner[i] < 0 else int(self.t_scale[owner[i]])
            if t > 0:                     # torch creates a parameter's state at its first update
                state[i] = {'step': torch.tensor(float(t)), 'exp_avg': self.exp_avg[off:off + n].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False,
                 'initial_lr': self.initial_lr, 'params': list(range(len(self.model._plist)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        self.lr = float(g['lr'])
        self.initial_lr = float(g.get('initial_lr', self.initial_lr))
        self.betas, self.eps, self.weight_decay = tuple(g['betas']), float(g['eps']), float(g['weight_decay'])
        off = 0
        steps = set()
        owner = self._scale_of_param()
        self.t_scale[:] = 0
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, p in enumerate(self.model._plist):
            n = p.numel()
            st = sd['state'].get(i)
            if st is not None:
                self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                t = int(float(st['step']))
                if owner[i] < 0:
                    steps.add(t)
                else:
                    if self.t_scale[owner[i]] not in (0, t):
                        raise ValueError('step counts differ inside scale_mlp.%d' % owner[i])
                    self.t_scale[owner[i]] = t
            off += n
        if len(steps) > 1:
            raise ValueError('step counts of the shared parameters differ; the fused Adam keeps one counter for them')
        self.t = steps.pop() if steps else 0


def train_step(model, opt, frame, point_num, out=None):
    """One iteration of main.py:305-321 on a batched frame: bits -> loss = bits/point_num -> backward -> Adam ->
    StepLR, as ONE C-ABI call (linr_net_train_step).  Returns the device-resident bits accumulator (float64[1]);
    nothing synchronises with the host.  `out`: a zeroed float64[1] device tensor to add the bits into (e.g. one slot of a
    per-GOP vector that is cleared once per epoch) - saves the per-step allocation + fill."""
    bits = torch.zeros(1, dtype=torch.float64, device=frame.device) if out is None else out
    if model._wide is not None:          # hidden_channel_conv 16 / 32: the channel-blocked executor + the segment-wise Adam
        if model.train_precision != 'f32':
            raise _lib.LinrError('the bf16 training executor exists for hidden_channel_conv=8 only')
        with torch.no_grad(), model._wide.lock:          # (the pooled buffers of the forward must survive until the backward has run)
            tape = model._wide.forward(frame, 0, 8, None, bits, keep=True, pool=True)
            model._ensure_grad_views()
            model._flat_grad.zero_()
            if tape is not None:
                model._wide.backward(frame, tape, 1.0 / float(point_num), pool=True)
            opt.grad.copy_(model._flat_grad)
            opt.step(frame)
        opt.scheduler_step()
        return bits
    t, t_scale = opt.advance(frame)
    if model.train_precision == 'bf16':
        if model.block_layers != 1:
            raise _lib.LinrError('the bf16 training executor supports block_layers=1 only')
        engine.net_train_step_bf16(frame, model.flat_parameters(), opt.exp_avg, opt.exp_avg_sq, 1.0 / float(point_num), t,
                                   opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.weight_decay, bits, scale_steps=t_scale)
    elif model.train_precision == 'f32':
        engine.net_train_step(frame, model.flat_parameters(), opt.exp_avg, opt.exp_avg_sq, 1.0 / float(point_num), t,
                              opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.weight_decay, bits, scale_steps=t_scale)
    else:
        raise ValueError("train_precision must be 'f32' or 'bf16'")
    opt.t, opt.t_scale = t, t_scale              # committed only after the call returned without an error
    opt.scheduler_step()
    return bits
