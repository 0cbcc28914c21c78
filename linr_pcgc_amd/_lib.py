"""ctypes binding of liblinr_hip.so (include/linr_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os

import torch  # noqa: F401  - MUST precede loading liblinr_hip.so: both link libamdhip64.so.7 and the process has to
#                             end up with ONE HIP runtime, the one PyTorch bundles (streams and memory come from it).

_HERE = os.path.dirname(os.path.abspath(__file__))
# LINR_HIP_LIB: another build of the same library (kernel experiments, tools/wgrad_floor_lab.sh); the ABI check below still applies
LIB_PATH = os.environ.get('LINR_HIP_LIB') or os.path.join(_HERE, 'liblinr_hip.so')

LINR_RELU, LINR_ACCUM, LINR_RELU_MASK, LINR_NO_BIAS, LINR_PAD_ROW = 1, 2, 4, 8, 16
LINR_FRAME_OCC_PADDED = 1
ABI_VERSION = 11

c_i32, c_i64, c_u32, c_f32, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float, ctypes.c_double
c_ptr, c_size = ctypes.c_void_p, ctypes.c_size_t


class LinrWidePw(ctypes.Structure):
    """linr_wide_pw of include/linr_hip.h: a pointwise layer fused into a wide convolution's epilogue."""
    _fields_ = [('mode', ctypes.c_int32), ('W', ctypes.c_void_p), ('b', ctypes.c_void_p), ('aux_h', ctypes.c_void_p), ('out2_h', ctypes.c_void_p)]


class LinrWideReduce(ctypes.Structure):
    """linr_wide_reduce of include/linr_hip.h: one deferred weight-gradient reduction."""
    _fields_ = [('kind', ctypes.c_int32), ('nblocks', ctypes.c_int32), ('cin', ctypes.c_int32), ('cout', ctypes.c_int32),
                ('ws_ci', ctypes.c_int32), ('ws_co', ctypes.c_int32), ('slab', ctypes.c_void_p), ('gW', ctypes.c_void_p),
                ('gb', ctypes.c_void_p)]


class LinrFrame(ctypes.Structure):
    """struct linr_frame (include/linr_hip.h)."""
    _fields_ = [('rows', c_i64), ('n_scales', c_i32), ('model_scale_num', c_i32), ('block_layers', c_i32),
                ('flags', c_i32), ('row_off_h', c_ptr),
                ('scale_idx_h', c_ptr), ('nbr', c_ptr), ('nbr_ld', c_i64), ('nbr_lo', c_ptr), ('nbr_mask', c_ptr),
                ('offset_feat', c_ptr), ('occ', c_ptr), ('nbr8t', c_ptr)]


class LinrInceptionParams(ctypes.Structure):
    """struct linr_inception_params (include/linr_hip.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ('w00', 'b00', 'w01', 'b01', 'w10', 'b10', 'w11', 'b11', 'w12', 'b12')]


_PROTOS = {
    'linr_abi_version': (ctypes.c_int, []),
    'linr_param_count': (c_i64, [c_i32, c_i32]),
    'linr_kmap_workspace_bytes': (c_size, [c_i64]),
    'linr_kmap_build': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_size, c_ptr]),
    'linr_kmap_validate': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    'linr_spconv_wgrad_cmap_blocks': (ctypes.c_int64, []),
    'linr_spconv_wgrad_cmap': (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr]),
    'linr_spconv_bwd_fused': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i32, c_ptr]),
    'linr_inception_bwd_fused': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                                                ctypes.POINTER(LinrInceptionParams), c_ptr, c_ptr, c_u32, c_ptr, c_i32, c_ptr]),
    'linr_slab_reduce': (ctypes.c_int, [c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_u32, c_ptr]),
    'linr_axpy': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i32, c_ptr]),
    'linr_sum_many': (ctypes.c_int, [c_ptr, c_i32, c_i64, c_ptr, c_i32, c_ptr]),
    'linr_occ_wgrad7': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i32, c_ptr, c_ptr]),
    'linr_kmap_tile8t_bytes': (c_size, [c_i64]),
    'linr_kmap_tile8t': (ctypes.c_int, [c_ptr, c_i64, c_i64, c_ptr, c_size, c_ptr]),
    'linr_prof_mask': (ctypes.c_int, [c_u32]),
    'linr_prof_enable': (ctypes.c_int, [c_i32]),
    'linr_debug_poison': (ctypes.c_int, [c_u32]),
    'linr_debug_poison_now': (ctypes.c_int, [c_ptr]),
    'linr_prof_read': (ctypes.c_int, [c_i32, c_ptr, c_ptr, c_ptr]),
    'linr_octree_occupancy': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    'linr_sort_unique_workspace_bytes': (ctypes.c_size_t, [c_i64]),
    'linr_coords_sort_unique': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    'linr_coords_minmax': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    'linr_octree_level': (ctypes.c_int, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    'linr_octree_levels_count': (c_i32, [c_i32, c_i32]),
    'linr_octree_levels_rows': (c_i64, [c_i64, c_i32, c_i32]),
    'linr_octree_levels_workspace_bytes': (ctypes.c_size_t, [c_i64, c_i32, c_i32]),
    'linr_octree_levels': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    'linr_kmap_offset_feat': (ctypes.c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'linr_kmap_compress': (ctypes.c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr]),
    'linr_spconv_fwd': (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i32,
                                       c_ptr, c_i32, c_u32, c_ptr]),
    'linr_spconv_bwd_data': (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_i64, c_i64, c_ptr, c_i32, c_i32, c_ptr, c_i32,
                                            c_ptr, c_i32, c_u32, c_ptr]),
    'linr_spconv_wide': (ctypes.c_int, [c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, ctypes.c_uint32, c_ptr]),
    'linr_spconv_wgrad_wide_slab_bytes': (ctypes.c_size_t, [c_i32, c_i32]),
    'linr_spconv_wgrad_wide': (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'linr_linear_wide': (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int32, c_ptr, ctypes.c_int32, ctypes.c_int32, c_ptr, ctypes.c_int32,
                                        ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_uint32, c_ptr]),
    'linr_spconv_wide_pw': (ctypes.c_int, [c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_u32,
                                           ctypes.POINTER(LinrWidePw), c_ptr]),
    'linr_spconv_wgrad_wide2': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'linr_spconv_wgrad_wide_blocks': (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]),
    'linr_linear_wgrad_wide_blocks': (ctypes.c_int32, [c_i64]),
    'linr_wide_reduce_many': (ctypes.c_int, [ctypes.POINTER(LinrWideReduce), ctypes.c_int32, c_ptr]),
    'linr_head_wide_workspace_bytes': (ctypes.c_size_t, [c_i64]),
    'linr_bits_finish': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    'linr_head_wide_fwd': (ctypes.c_int, [c_ptr, ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int32, c_i64, c_ptr, c_ptr, c_ptr,
                                          ctypes.c_size_t, c_ptr]),
    'linr_head_wide_bwd_slab_bytes': (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    'linr_head_wide_bwd': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_float, c_ptr, c_i64, c_ptr, ctypes.c_size_t, c_ptr, c_ptr]),
    'linr_linear_wgrad_wide_workspace_bytes': (ctypes.c_size_t, [c_i64, ctypes.c_int32, ctypes.c_int32]),
    'linr_linear_wgrad_wide': (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int32, c_ptr, ctypes.c_int32, ctypes.c_int32, c_i64, c_ptr,
                                              ctypes.c_int32, ctypes.c_int32, c_ptr, ctypes.c_uint32, c_ptr, ctypes.c_size_t, c_ptr]),
    'linr_spconv_cmap': (ctypes.c_int, [c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_i32, c_ptr,
                                        c_i32, c_ptr, c_i32, c_ptr, c_i32, c_u32, c_ptr]),
    'linr_spconv_bwd_weight_workspace_bytes': (c_size, [c_i64, c_i32, c_i32]),
    'linr_spconv_bwd_weight': (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr,
                                              c_ptr, c_u32, c_ptr, c_size, c_ptr]),
    'linr_linear_fwd': (ctypes.c_int, [c_ptr, c_i32, c_i64, c_ptr, c_i32, c_i32, c_ptr, c_i32, c_i32, c_ptr, c_i32,
                                       c_ptr, c_i32, c_u32, c_ptr]),
    'linr_linear_bwd_data': (ctypes.c_int, [c_ptr, c_i32, c_i64, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i32,
                                            c_ptr, c_i32, c_u32, c_ptr]),
    'linr_linear_bwd_weight_workspace_bytes': (c_size, [c_i64, c_i32, c_i32]),
    'linr_linear_bwd_weight': (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_i32, c_i64, c_i32, c_i32, c_ptr, c_i32, c_i32,
                                              c_ptr, c_u32, c_ptr, c_size, c_ptr]),
    'linr_bce_workspace_bytes': (c_size, [c_i64]),
    'linr_bce_bits_fwd': (ctypes.c_int, [c_ptr, c_ptr, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'linr_bce_bits_bwd': (ctypes.c_int, [c_ptr, c_ptr, c_i32, c_i64, c_f32, c_ptr, c_ptr]),
    'linr_adam_step': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f64, c_f64, c_f64, c_f64, c_f64, c_f64,
                                      c_ptr]),
    'linr_net_arena_bytes': (c_size, [c_i64, c_i32]),
    'linr_net_forward': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_size, c_i32, c_i32, c_ptr, c_ptr,
                                        c_ptr]),
    'linr_net_backward': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_size, c_f32, c_ptr, c_ptr]),
    'linr_net_train_step': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_size, c_f32, c_ptr, c_ptr, c_f64,
                                           c_i64, c_ptr, c_f64, c_f64, c_f64, c_f64, c_ptr, c_ptr]),
    'linr_net_bf16_arena_bytes': (c_size, [c_i64, c_i32]),
    'linr_net_forward_bf16': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_f32, c_f32, c_ptr, c_size, c_i32, c_i32, c_ptr,
                                             c_ptr, c_ptr]),
    'linr_net_train_bf16_arena_bytes': (c_size, [c_i64, c_i32]),
    'linr_occ_to_bf16': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    'linr_net_forward_train_bf16': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_size, c_ptr, c_ptr, c_ptr, c_ptr]),
    'linr_net_backward_bf16': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_size, c_ptr, c_f32, c_ptr, c_ptr]),
    'linr_net_train_step_bf16': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_size, c_ptr, c_f32, c_ptr, c_ptr, c_f64,
                                                c_i64, c_ptr, c_f64, c_f64, c_f64, c_f64, c_ptr, c_ptr]),
    'linr_spconv_bwd_fused_bf16': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr]),
    'linr_sce_fwd': (ctypes.c_int, [c_ptr, ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_ptr, c_ptr]),
    'linr_sce_bwd': (ctypes.c_int, [c_ptr, ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_ptr, c_ptr]),
    'linr_sce_param_count': (ctypes.c_int64, [ctypes.c_int32]),
    'linr_sce_bwd_params_slab_bytes': (ctypes.c_size_t, [ctypes.c_int32]),
    'linr_sce_bwd_params': (ctypes.c_int, [c_ptr, ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr, c_ptr]),
    'linr_head_workspace_bytes': (c_size, [c_i64]),
    'linr_head_fwd': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32,
                                     c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'linr_head_bwd': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_i64, c_ptr, c_ptr, c_size,
                                     c_ptr]),
    'linr_inception_fwd': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, ctypes.POINTER(LinrInceptionParams), c_ptr, c_ptr,
                                          c_ptr, c_ptr]),
    'linr_inception_bwd_data': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                                               ctypes.POINTER(LinrInceptionParams), c_ptr, c_ptr, c_ptr, c_u32, c_ptr]),
    'linr_spconv_wgrad_dual44': (ctypes.c_int, [c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'linr_occ_conv7': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'linr_ac_encode_binary': (c_i64, [c_ptr, c_ptr, c_i64, c_ptr, c_i64]),
    'linr_ac_decode_binary': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'linr_net_decode_stages': (ctypes.c_int, [ctypes.POINTER(LinrFrame), c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_size, c_ptr, c_ptr, c_ptr,
                                              c_ptr, c_ptr, c_ptr, c_ptr]),
    'linr_decode_scale_ws_bytes': (c_size, [c_i64, c_i32, c_i32]),
    'linr_decode_scale': (ctypes.c_int, [c_ptr, c_i64, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_size,
                                         c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'linr_ac_encode_cdf16': (c_i64, [c_ptr, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_i64]),
    'linr_ac_decode_cdf16': (ctypes.c_int, [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_i64, c_ptr]),
    'linr_ply_parse_ascii': (ctypes.c_int, [c_ptr, c_size, c_i64, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr]),
    'linr_ac_encode_binary_batch': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32]),
}

EXPORTS = tuple(_PROTOS)
_lib = None


class LinrError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises (never falls back) when liblinr_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LinrError('%s not found: build it with linr_pcgc_amd/csrc/build.sh (or __graft_entry__.build()); '
                            'there is no CPU / PyTorch fallback for the coding network' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)          # AttributeError if an export is missing
            fn.restype, fn.argtypes = res, args
        if handle.linr_abi_version() != ABI_VERSION:
            raise LinrError('liblinr_hip.so ABI %d != binding ABI %d' % (handle.linr_abi_version(), ABI_VERSION))
        if os.environ.get('LINR_DEBUG_POISON'):
            handle = _Poisoned(handle)
        _lib = handle
    return _lib


class _Poisoned:
    """LINR_DEBUG_POISON=1 (test aid, tools/README.md): every GPU entry point is called with the LDS and vector registers of all
    CUs freshly filled with a NaN pattern (linr_debug_poison_now in front of it, linr_debug_poison(all classes) for the launches
    inside the executors).  The test suite must pass unchanged under it: no kernel may read on-chip state it did not write."""
    _HOST = ('linr_ac_', 'linr_prof_', 'linr_debug_', 'linr_abi_version')

    def __init__(self, handle):
        self._h = handle
        handle.linr_debug_poison(0xFFFFFF)

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name.startswith(self._HOST) or name.endswith('_bytes'):
            return fn
        h = self._h

        def call(*a):
            import torch
            if torch.cuda.is_available():
                h.linr_debug_poison_now(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            return fn(*a)
        return call


_ERR = {-1: 'LINR_EINVAL (bad argument)', -2: 'LINR_ENOSPC (buffer too small)', -3: 'LINR_EALIGN (misaligned pointer)'}


def check(rc, what):
    """The reference's only error convention is Python assert/ValueError (encoder.py:86-87); we raise RuntimeError."""
    if rc != 0:
        raise LinrError('%s failed: %s' % (what, _ERR.get(rc, 'hipError_t %d' % rc)))


def scratch(nbytes, device):
    """Uninitialised device scratch (arenas, workspaces, slabs) that the library must initialise itself wherever it reads it.
    Under LINR_DEBUG_POISON it comes filled with 0xFF bytes (NaN as float, -1 as int) instead of whatever the allocator returns -
    mostly zeros from a fresh process, which is exactly what hides a missing initialisation."""
    import torch
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if os.environ.get('LINR_DEBUG_POISON'):
        t.fill_(0xFF)
    return t


def current_stream_handle():
    """The raw handle of PyTorch's current HIP stream on the current device.  torch.cuda.current_stream() builds a Stream object through
    several Python layers (~9 us per call; a frame's staging makes twenty such calls): the two C entry points below are what it ends in."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:                                         # (a torch build without them)
        return torch.cuda.current_stream().cuda_stream
