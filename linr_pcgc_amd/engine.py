"""Frame descriptors and whole-network calls (linr_net_forward / linr_net_backward of include/linr_hip.h).

A ``Frame`` is what the reference passes scale by scale to ``LINR_PCGC_Model.forward`` (main.py:457-475): here all
scales of one point cloud share one row space (finest first) so the whole multiscale network is one launch sequence.
Coordinates are static over all epochs, so the kernel map is built once per frame and cached.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from ._lib import check


class Frame:
    """All scales of one frame on the GPU: kernel map, 7-neighbour features, child occupancy, activation arena."""

    def __init__(self, scales, model_scale_num, device='cuda', validate=True, with_arena=True, block_layers=1):
        """scales: list of dicts {'coord' int32 [N,3] sorted x-major, 'offset_tensor' [N,7] float (None: derived from the
        kernel map),
        'occ' [N,8] float or 'occ_lst' 8 x [N,1], 'scale_idx'} - the per-scale network inputs of
        datautils/custom_dataset.py:318-323.  Missing 'occ' (decoder) gives a zero occupancy buffer."""
        device = torch.device(device)
        if device.type != 'cuda':
            raise _lib.LinrError('Frame needs a GPU device: the coding network has no CPU path')
        self.device = device
        self.model_scale_num = int(model_scale_num)
        self.block_layers = int(block_layers)
        for i, s in enumerate(scales):          # shapes first: the per-scale dicts come from user code (the drivers' putin_args)
            c = s['coord']
            n = int(c.shape[0])
            if c.ndim != 2 or c.shape[1] != 3:
                raise ValueError("scale %d: 'coord' must be [N, 3], got %s" % (i, tuple(c.shape)))
            if not 0 <= int(s['scale_idx']) < self.model_scale_num:
                raise ValueError("scale %d: 'scale_idx' %d is outside the model's 0..%d" % (i, int(s['scale_idx']), self.model_scale_num - 1))
            off = s.get('offset_tensor')
            if off is not None and tuple(off.shape) != (n, 7):
                raise ValueError("scale %d: 'offset_tensor' must be [%d, 7], got %s" % (i, n, tuple(off.shape)))
            if s.get('occ') is not None:
                if tuple(s['occ'].shape) != (n, 8):
                    raise ValueError("scale %d: 'occ' must be [%d, 8], got %s" % (i, n, tuple(s['occ'].shape)))
            elif s.get('occ_lst') is not None:
                lst = s['occ_lst']
                if len(lst) != 8 or any(int(torch.as_tensor(o).numel()) != n for o in lst):
                    raise ValueError("scale %d: 'occ_lst' must hold 8 columns of %d values, got %d entries of sizes %s"
                                     % (i, n, len(lst), sorted({int(torch.as_tensor(o).numel()) for o in lst})))
        ns = [int(s['coord'].shape[0]) for s in scales]
        self.row_off = np.zeros(len(scales) + 1, dtype=np.int64)
        self.row_off[1:] = np.cumsum(ns)
        self.rows = int(self.row_off[-1])
        self.scale_idx = np.asarray([int(s['scale_idx']) for s in scales], dtype=np.int32)
        self.n_scales = len(scales)
        R = self.rows
        self.nbr_ld = (R + 63) // 64 * 64          # padded leading dimension: 16-byte aligned index rows
        # Buffers are allocated uninitialised and only what the kernels below do not write is filled (the padding columns of the maps,
        # the zero row in front of the occupancy): a 337 k-row frame's neighbour table alone is 36 MB.
        self.nbr = torch.empty((27, self.nbr_ld), dtype=torch.int32, device=device)
        if self.nbr_ld > R:
            self.nbr[:, R:] = -1                                                           # padding columns: no neighbour
        self.offset_feat = torch.empty((R, 7), dtype=torch.float32, device=device)
        # occupancy with a zero row in front: the executor gathers it in place (LINR_FRAME_OCC_PADDED)
        self._occ_buf = torch.empty((R + 1, 8), dtype=torch.float32, device=device)
        self._occ_buf[0].zero_()
        self.occ = self._occ_buf[1:]
        bad = torch.zeros(1, dtype=torch.int32, device=device) if validate else None       # ONE flag for all scales, read once below
        have_occ = [('occ' in s and s['occ'] is not None) or ('occ_lst' in s and s['occ_lst'] is not None) for s in scales]
        if not all(have_occ):
            self.occ.zero_()                                                               # decoder: the occupancy is filled stage by stage
        need_feat = []
        L = _lib.lib()
        for i, s in enumerate(scales):
            r0, r1 = int(self.row_off[i]), int(self.row_off[i + 1])
            if r1 == r0:
                continue
            coord = torch.as_tensor(s['coord']).to(device=device, dtype=torch.int32).contiguous()
            if validate:
                check(L.linr_kmap_validate(coord.data_ptr(), r1 - r0, bad.data_ptr(), _stream()), 'linr_kmap_validate')
            ops.kmap_build_into(coord, self.nbr, r0, False)
            if s.get('offset_tensor') is None:       # decoder fast path: the 7-neighbour occupancy is part of the kernel map
                need_feat.append((r0, r1))
            else:
                self.offset_feat[r0:r1] = torch.as_tensor(s['offset_tensor']).to(device=device, dtype=torch.float32)
            if 'occ' in s and s['occ'] is not None:
                self.occ[r0:r1] = torch.as_tensor(s['occ']).to(device=device, dtype=torch.float32)
            elif 'occ_lst' in s and s['occ_lst'] is not None:
                self.occ[r0:r1] = torch.cat([torch.as_tensor(o).reshape(-1, 1) for o in s['occ_lst']], dim=1).to(
                    device=device, dtype=torch.float32)
        # the kernel map holds GLOBAL row ids, so adjacent row ranges share one launch (all scales of a staged frame: one)
        merged = []
        for r0, r1 in need_feat:
            if merged and merged[-1][1] == r0:
                merged[-1][1] = r1
            else:
                merged.append([r0, r1])
        for r0, r1 in merged:
            check(L.linr_kmap_offset_feat(self.nbr.data_ptr(), self.nbr_ld, r0, r1 - r0, self.offset_feat[r0:r1].data_ptr(), _stream()),
                  'linr_kmap_offset_feat')
        # compressed kernel map (9 column bases + 27-bit mask per row) used by the network executor
        self.nbr_lo = torch.empty((9, self.nbr_ld), dtype=torch.int32, device=device)
        self.nbr_mask = torch.empty((self.nbr_ld,), dtype=torch.int32, device=device)
        if self.nbr_ld > R:
            self.nbr_lo[:, R:] = 0
            self.nbr_mask[R:] = 0                                                          # padding columns: empty masks
        check(_lib.lib().linr_kmap_compress(self.nbr.data_ptr(), self.nbr_ld, R, self.nbr_lo.data_ptr(),
                                            self.nbr_mask.data_ptr(), self.nbr_ld, _stream()), 'linr_kmap_compress')
        # the kernel map tiled by 8 rows in the gather-lane order of the stand-alone weight-gradient kernels (128 B per row)
        nb8t = _lib.lib().linr_kmap_tile8t_bytes(R)
        self.nbr8t = torch.empty(max(4, (nb8t + 3) // 4), dtype=torch.int32, device=device)
        check(_lib.lib().linr_kmap_tile8t(self.nbr.data_ptr(), self.nbr_ld, R, self.nbr8t.data_ptr(), self.nbr8t.numel() * 4,
                                          _stream()), 'linr_kmap_tile8t')
        if validate and int(bad.item()) != 0:                                              # the frame's one host read
            raise ValueError('coord must be unique, non-negative (< 2^20) and sorted by the x-major ravel key '
                             '(models/module_utils.py:246-256)')
        self.arena = None
        if with_arena:
            self.alloc_arena()
        self._c = _lib.LinrFrame(rows=R, n_scales=self.n_scales, model_scale_num=self.model_scale_num,
                                 block_layers=self.block_layers, flags=_lib.LINR_FRAME_OCC_PADDED,
                                 row_off_h=self.row_off.ctypes.data, scale_idx_h=self.scale_idx.ctypes.data,
                                 nbr=self.nbr.data_ptr(), nbr_ld=self.nbr_ld, nbr_lo=self.nbr_lo.data_ptr(),
                                 nbr_mask=self.nbr_mask.data_ptr(), offset_feat=self.offset_feat.data_ptr(),
                                 occ=self.occ.data_ptr(), nbr8t=self.nbr8t.data_ptr())

    def alloc_arena(self):
        nbytes = _lib.lib().linr_net_arena_bytes(self.rows, self.block_layers)
        self.arena = _lib.scratch(nbytes, self.device)

    def bf16_arena(self):
        """Arena of the bf16 / uint8-weight inference executor (allocated on first use, 64-byte aligned)."""
        if getattr(self, '_arena_bf16', None) is None:
            nbytes = _lib.lib().linr_net_bf16_arena_bytes(self.rows, self.block_layers)
            self._arena_bf16 = _lib.scratch(nbytes + 64, self.device)
        return self._arena_bf16

    def train_bf16_arena(self):
        """Arena of the bf16 training executor (linr_net_train_step_bf16): the GOP's shared one when a Gop set it, else this frame's
        own (allocated on first use).  Returns (64-byte aligned base address, usable bytes)."""
        if getattr(self, 'arena_train_bf16', None) is None:
            nbytes = _lib.lib().linr_net_train_bf16_arena_bytes(self.rows, self.block_layers)
            if nbytes == 0:
                raise _lib.LinrError('the bf16 training executor supports block_layers=1 only')
            self.arena_train_bf16 = _lib.scratch(nbytes + 64, self.device)
        a = self.arena_train_bf16
        base = (a.data_ptr() + 63) & ~63
        return base, a.numel() - (base - a.data_ptr())

    def invalidate_occ(self):
        """To be called by whoever rewrites frame.occ in place (the decode paths): drops the cached bf16 copy, so that a later bf16
        forward / training step converts the NEW occupancy instead of mixing old inputs with new targets."""
        self._occ_bf16 = None

    def occ_bf16(self):
        """The frame's occupancy as bf16 [1 + rows][8] with the zero row in front, converted once (linr_occ_to_bf16): it does not
        change over the epochs of an overfit (writers of frame.occ call invalidate_occ()).  Returns the address of the ZERO row."""
        if getattr(self, '_occ_bf16', None) is None:
            t = torch.empty((self.rows + 1, 8), dtype=torch.int16, device=self.device)
            check(_lib.lib().linr_occ_to_bf16(self.occ.data_ptr(), self.rows, t.data_ptr(), _stream()), 'linr_occ_to_bf16')
            self._occ_bf16 = t
        return self._occ_bf16.data_ptr()

    def cref(self):
        return ctypes.byref(self._c)

    def scale_slice(self, i):
        return slice(int(self.row_off[i]), int(self.row_off[i + 1]))


def _stream():
    return _lib.current_stream_handle()


def net_forward(frame, flat_params, stage_begin=0, stage_end=8, probs=None, bits=None, arena=None):
    """Runs stages [stage_begin, stage_end).  probs: float32 [8, rows] or None; bits: float64[1] accumulator or None."""
    arena = frame.arena if arena is None else arena
    check(_lib.lib().linr_net_forward(frame.cref(), flat_params.data_ptr(), arena.data_ptr(), arena.numel(), stage_begin,
                                      stage_end, 0 if probs is None else probs.data_ptr(),
                                      0 if bits is None else bits.data_ptr(), _stream()), 'linr_net_forward')


def net_forward_bf16(frame, codes, min_param, max_param, stage_begin, stage_end, probs, bits=None):
    """linr_net_forward_bf16: inference from the uint8 weight codes with bf16 features.  probs: float32 [8, rows]."""
    arena = frame.bf16_arena()
    base = (arena.data_ptr() + 63) & ~63
    check(_lib.lib().linr_net_forward_bf16(frame.cref(), codes.data_ptr(), float(min_param), float(max_param), base,
                                           arena.numel() - (base - arena.data_ptr()), stage_begin, stage_end, probs.data_ptr(),
                                           0 if bits is None else bits.data_ptr(), _stream()), 'linr_net_forward_bf16')


def net_decode_stages(frame, flat_params, streams_per_scale, probs, p_pinned, s_pinned, s_dev, qcodes=None, qrange=None):
    """linr_net_decode_stages: the 8 decode stages of a frame object (stage forward, D2H, range decoder, H2D, occupancy column)
    in one C call that does not hold the GIL.  streams_per_scale: [n_scales][8] byte strings; qcodes / qrange select the bf16 /
    uint8-weight executor.  frame.occ must be zeroed by the caller and holds the decoded occupancy afterwards."""
    n = frame.n_scales
    bufs = [np.frombuffer(streams_per_scale[i][k], dtype=np.uint8) for i in range(n) for k in range(8)]
    ptrs = (ctypes.c_void_p * (8 * n))(*[b.ctypes.data if b.size else None for b in bufs])
    lens = (ctypes.c_int64 * (8 * n))(*[int(b.size) for b in bufs])
    if qcodes is None:
        arena, codes, lo, hi = frame.arena, None, 0.0, 0.0
        base, nbytes = arena.data_ptr(), arena.numel()
    else:
        arena = frame.bf16_arena()
        base = (arena.data_ptr() + 63) & ~63
        nbytes = arena.numel() - (base - arena.data_ptr())
        codes, lo, hi = qcodes.data_ptr(), float(qrange[0]), float(qrange[1])
    frame.invalidate_occ()                         # the call rewrites frame.occ column by column
    check(_lib.lib().linr_net_decode_stages(frame.cref(), None if flat_params is None else flat_params.data_ptr(), codes, lo, hi,
                                            base, nbytes, ptrs, lens, probs.data_ptr(), p_pinned.data_ptr(), s_pinned.data_ptr(),
                                            s_dev.data_ptr(), _stream()), 'linr_net_decode_stages')


def net_backward(frame, flat_params, flat_grads, gscale, arena=None):
    """flat_grads += gscale * d bits / d params (needs a preceding full net_forward on the same arena)."""
    arena = frame.arena if arena is None else arena
    check(_lib.lib().linr_net_backward(frame.cref(), flat_params.data_ptr(), arena.data_ptr(), arena.numel(), float(gscale),
                                       flat_grads.data_ptr(), _stream()), 'linr_net_backward')


def net_train_step(frame, flat_params, exp_avg, exp_avg_sq, gscale, step, lr, beta1, beta2, eps, weight_decay, bits,
                   arena=None, scale_steps=None):
    """linr_net_train_step: forward + backward + deterministic gradient reduction + fused Adam in one call.
    scale_steps: int64 numpy array [model_scale_num] of the per-scale context MLPs' 1-based step counts (torch.optim.Adam
    semantics for scales a frame does not contain), or None = every parameter uses `step`."""
    arena = frame.arena if arena is None else arena
    check(_lib.lib().linr_net_train_step(frame.cref(), flat_params.data_ptr(), arena.data_ptr(), arena.numel(),
                                         float(gscale), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), float(lr), int(step),
                                         None if scale_steps is None else scale_steps.ctypes.data,
                                         beta1, beta2, eps, weight_decay, bits.data_ptr(), _stream()),
          'linr_net_train_step')


def net_forward_train_bf16(frame, flat_params, probs=None, bits=None):
    """linr_net_forward_train_bf16: the teacher-forced forward of the bf16 training executor (activations kept in its arena)."""
    base, nbytes = frame.train_bf16_arena()
    check(_lib.lib().linr_net_forward_train_bf16(frame.cref(), flat_params.data_ptr(), base, nbytes, frame.occ_bf16(),
                                                 0 if probs is None else probs.data_ptr(), 0 if bits is None else bits.data_ptr(),
                                                 _stream()), 'linr_net_forward_train_bf16')


def net_backward_bf16(frame, flat_params, flat_grads, gscale):
    """flat_grads += gscale * d bits / d params through the bf16 training executor (needs net_forward_train_bf16 before it)."""
    base, nbytes = frame.train_bf16_arena()
    check(_lib.lib().linr_net_backward_bf16(frame.cref(), flat_params.data_ptr(), base, nbytes, frame.occ_bf16(), float(gscale),
                                            flat_grads.data_ptr(), _stream()), 'linr_net_backward_bf16')


def net_train_step_bf16(frame, flat_params, exp_avg, exp_avg_sq, gscale, step, lr, beta1, beta2, eps, weight_decay, bits,
                        scale_steps=None):
    """linr_net_train_step_bf16: net_train_step with bf16 feature / gradient rows, fp32 master parameters and accumulation."""
    base, nbytes = frame.train_bf16_arena()
    check(_lib.lib().linr_net_train_step_bf16(frame.cref(), flat_params.data_ptr(), base, nbytes, frame.occ_bf16(),
                                              float(gscale), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), float(lr), int(step),
                                              None if scale_steps is None else scale_steps.ctypes.data,
                                              beta1, beta2, eps, weight_decay, bits.data_ptr(), _stream()),
          'linr_net_train_step_bf16')
