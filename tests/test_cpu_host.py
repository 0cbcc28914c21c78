"""CPU: host logic, the C-ABI library's exports, the C++ range coder against the oracle, golden-vector checks of the
product's own octree prep and bitstream container.  No GPU compute is called here."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import ac as oac
from oracle import model_codec as omc
from oracle import octree as ooct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from linr_pcgc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    from linr_pcgc_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'linr_hip.h')).read()
    declared = set(re.findall(r'^LINR_API\s+[\w\s\*]+?\b(linr_\w+)\(', header, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.linr_abi_version() == _lib.ABI_VERSION


def test_c_abi_argument_checks(lib):
    """Every entry validates its arguments before it touches the device: bad sizes / NULLs / short or misaligned workspaces
    come back as LINR_EINVAL (-1) / LINR_ENOSPC (-2) / LINR_EALIGN (-3), with nothing launched (so this runs without a GPU)."""
    import ctypes
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    # kernel map
    assert lib.linr_kmap_build(p16, -1, p16, 8, 0, p16, 1024, None) == -1                  # n < 0
    assert lib.linr_kmap_build(p16, 8, p16, 4, 0, p16, 1024, None) == -1                   # ld < n
    assert lib.linr_kmap_build(p16, 8, p16, 8, 4, p16, 1024, None) == -1                   # row_base + n > ld
    assert lib.linr_kmap_build(None, 8, p16, 8, 0, p16, 1024, None) == -1                  # NULL coords
    assert lib.linr_kmap_build(p16, 8, p16, 8, 0, p16, 8, None) == -2                      # workspace too small
    assert lib.linr_kmap_build(p16, 8, p16, 8, 0, p16 + 4, 1024, None) == -3               # workspace not 8-byte aligned
    assert lib.linr_kmap_build(p16, 0, p16, 0, 0, None, 0, None) == 0                      # empty input is fine
    assert lib.linr_kmap_offset_feat(p16, 4, 0, 8, p16, None) == -1
    assert lib.linr_octree_occupancy(p16, 8, p16, 4, p16, p16, 8, None) == -2                # workspace too small
    assert lib.linr_octree_occupancy(p16, -1, p16, 4, p16, p16, 1024, None) == -1
    # convolutions
    assert lib.linr_spconv_fwd(p16, 4, p16, 16, 16, p16, p16, 8, 8, None, 0, p16, 8, 0, None) == -1      # in_ld < cin
    assert lib.linr_spconv_cmap(0, p16, 5, p16, p16, 16, 16, p16, p16, 8, 8, None, 0, None, 0, p16, 8, 0, None) == -1   # ld 5
    assert lib.linr_spconv_cmap(0, p16 + 4, 8, p16, p16, 16, 16, p16, p16, 8, 8, None, 0, None, 0, p16, 8, 0, None) == -3
    assert lib.linr_spconv_wgrad_cmap(p16, 8, p16, 8, p16, None, 16, 16, 8, 5, p16, None) == -1              # cout 5
    assert lib.linr_spconv_bwd_fused(p16, p16, p16, p16, 16, 16, p16, p16 + 4, p16, 32, None) == -3          # gin not 16-byte aligned
    assert lib.linr_spconv_bwd_fused(p16, p16, p16, p16, 16, 16, p16, p16, p16, 0, None) == -1               # no slab rows
    assert lib.linr_spconv_bwd_weight(p16, 8, p16, 8, p16, 16, 16, 8, 8, p16, p16, 0, p16, 16, None) == -2             # ws short
    rows = ctypes.c_int32(7)
    g7 = (ctypes.c_void_p * 7)(*([p16] * 7))
    assert lib.linr_occ_wgrad7(p16, g7, p16, p16, 16, 16, p16, 0, ctypes.byref(rows), None) == -1            # no slab rows
    g7[3] = p16 + 4
    assert lib.linr_occ_wgrad7(p16, g7, p16, p16, 16, 16, p16, 8, ctypes.byref(rows), None) == -3            # a gradient matrix misaligned
    g7[3] = None
    assert lib.linr_occ_wgrad7(p16, g7, p16, p16, 16, 16, p16, 8, ctypes.byref(rows), None) == -1 and rows.value == 0
    assert lib.linr_occ_wgrad7(p16, g7, p16, p16, 16, 0, p16, 8, ctypes.byref(rows), None) == 0               # empty input is fine
    # round-4 entries: octree levels without torch.unique, convolutions on channel-blocked activations (widths 16 / 32)
    assert lib.linr_coords_sort_unique(p16, -1, None, 0, 20, p16, p16, p16, 1024, None) == -1               # n < 0
    assert lib.linr_coords_sort_unique(p16, 8, None, 0, 21, p16, p16, p16, 1024, None) == -1                # more than 20 bits
    assert lib.linr_coords_sort_unique(p16, 8, None, 20, 20, p16, p16, p16, 1024, None) == -1               # shift out of range
    assert lib.linr_coords_sort_unique(p16, 8, None, 0, 10, p16, None, p16, 1024, None) == -1               # no count
    assert lib.linr_octree_level(p16, -1, 10, p16, p16, p16, p16, 1024, None) == -1
    assert lib.linr_octree_level(p16, 8, 0, p16, p16, p16, p16, 1024, None) == -1                           # coord_bits < 1
    assert lib.linr_octree_level(p16, 8, 10, p16, p16, None, p16, 1024, None) == -1                         # no count
    assert lib.linr_coords_minmax(p16, 0, p16, None) == -1 and lib.linr_coords_minmax(p16, 8, None, None) == -1
    b2 = (ctypes.c_void_p * 2)(p16, p16)
    assert lib.linr_spconv_wide(0, b2, p16, p16, 16, 16, p16, p16, 16, 12, None, None, b2, 0, None) == -1   # cout not a multiple of 8
    assert lib.linr_spconv_wide(0, b2, p16, p16, 16, 16, p16, p16, 40, 16, None, None, b2, 0, None) == -1   # cin > 32
    assert lib.linr_spconv_wide(0, None, p16, p16, 16, 16, p16, p16, 16, 16, None, None, b2, 0, None) == -1
    assert lib.linr_spconv_wide(0, b2, p16, p16, 16, 16, p16, p16, 16, 16, None, None, b2, 64, None) == -1  # unknown flag
    assert lib.linr_spconv_wide(1, b2, p16, p16, 16, 16, p16, p16, 4, 16, None, None, b2, 0, None) == -1    # backward needs cin % 8 == 0
    assert lib.linr_spconv_wide(0, b2, p16, p16, 16, 16, p16, p16, 16, 16, None, None, b2, 4, None) == -1   # LINR_RELU_MASK without act blocks
    b2m = (ctypes.c_void_p * 2)(p16, p16 + 4)
    assert lib.linr_spconv_wide(0, b2m, p16, p16, 16, 16, p16, p16, 16, 16, None, None, b2, 0, None) == -3  # a gathered block misaligned
    assert lib.linr_spconv_wide(0, b2, p16, p16, 16, 0, p16, p16, 16, 16, None, None, b2, 0, None) == 0     # empty input is fine
    assert lib.linr_spconv_wgrad_wide(b2, 16, b2, 5, p16, None, 16, 16, p16, p16, p16, None) == -1          # cout 5
    assert lib.linr_spconv_wgrad_wide(b2, 16, b2, 16, p16, None, 16, 16, None, p16, p16, None) == -1        # no slab
    assert lib.linr_spconv_wgrad_wide_slab_bytes(16, 16) == 512 * 4 * 1736 * 4 and lib.linr_spconv_wgrad_wide_slab_bytes(0, 16) == 0
    # whole network: NULL frame / parameters, stage range
    assert lib.linr_net_forward(None, p16, p16, 4096, 0, 8, None, None, None) == -1
    assert lib.linr_net_train_step(None, p16, p16, 4096, 1.0, None, None, 0.01, 1, None, 0.9, 0.999, 1e-8, 1e-4, None, None) == -1
    assert lib.linr_param_count(0, 1) < 0
    assert lib.linr_param_count(7, 1) == 54712
    assert lib.linr_net_arena_bytes(-1, 1) == 0 and lib.linr_net_arena_bytes(100, 9) == 0
    assert lib.linr_net_arena_bytes(100, 2) > lib.linr_net_arena_bytes(100, 1) > 0
    # range coder: capacity / NULL checks
    assert lib.linr_ac_encode_binary(None, None, 4, p16, 64) == -1
    assert lib.linr_ac_decode_binary(None, 4, p16, 8, p16) == -1
    tot, nl, npass = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.linr_prof_read(24, ctypes.byref(tot), ctypes.byref(nl), ctypes.byref(npass)) == -1
    # bf16 training executor: NULL frame / arena, unsupported depth
    assert lib.linr_net_train_step_bf16(None, p16, p16, 4096, None, 1.0, None, None, 0.01, 1, None, 0.9, 0.999, 1e-8, 1e-4, None, None) == -1
    assert lib.linr_net_forward_train_bf16(None, p16, p16, 4096, None, None, None, None) == -1
    assert lib.linr_net_backward_bf16(None, p16, p16, 4096, None, 1.0, None, None) == -1
    assert lib.linr_net_train_bf16_arena_bytes(-1, 1) == 0 and lib.linr_net_train_bf16_arena_bytes(100, 2) == 0
    assert lib.linr_net_train_bf16_arena_bytes(100, 1) > 0
    assert lib.linr_occ_to_bf16(None, 4, None, None) == -1
    rw = ctypes.c_int32(7)
    assert lib.linr_spconv_bwd_fused_bf16(None, None, None, None, 4, 4, None, None, None, 1, ctypes.byref(rw), None) == -1 and rw.value == 0
    # round-4 entries on blocked activations: pointwise layers, their weight gradients, the occupancy head, the scale context's backward
    b4 = (ctypes.c_void_p * 4)(p16, p16, p16, p16)
    from linr_pcgc_amd._lib import LinrWidePw
    pw = LinrWidePw(1, p16, p16, None, None)
    assert lib.linr_spconv_wide_pw(0, b4, p16, p16, 16, 16, p16, p16, 16, 8, None, None, b4, 1, ctypes.byref(pw), None) == -1      # mode 1 without out2
    pw = LinrWidePw(3, p16, None, ctypes.cast(b4, ctypes.c_void_p), ctypes.cast(b4, ctypes.c_void_p))
    assert lib.linr_spconv_wide_pw(0, b4, p16, p16, 16, 16, p16, p16, 16, 16, None, None, b4, 0, ctypes.byref(pw), None) == -1     # mode 3 is a backward epilogue
    pw = LinrWidePw(2, p16, p16, ctypes.cast(b4, ctypes.c_void_p), ctypes.cast(b4, ctypes.c_void_p))
    assert lib.linr_spconv_wide_pw(0, b4, p16, p16, 16, 16, p16, p16, 16, 8, None, None, b4, 1, ctypes.byref(pw), None) == -1      # mode 2 needs h -> h
    assert lib.linr_linear_wide(b4, 12, 1, p16, 8, 1, p16, 8, 1, None, None, b4, 8, 0, None) == -1            # blocked cin not a multiple of 8
    assert lib.linr_linear_wide(b4, 16, 1, p16, 7, 1, p16, 8, 1, None, None, b4, 8, 0, None) == -1            # strides of no dense layout
    assert lib.linr_linear_wide(b4, 16, 1, p16, 8, 1, p16, 8, 1, None, None, b4, -1, 0, None) == -1           # n < 0
    assert lib.linr_linear_wide(None, 16, 1, p16, 8, 1, p16, 8, 1, None, None, b4, 8, 0, None) == -1
    assert lib.linr_linear_wide(b4, 16, 1, p16, 8, 1, p16, 8, 1, None, None, b4, 8, 4, None) == -1            # LINR_RELU_MASK without act
    assert lib.linr_linear_wide(b4, 16, 1, p16, 8, 1, p16, 8, 1, None, None, b4, 0, 0, None) == 0             # empty input is fine
    assert lib.linr_linear_wgrad_wide(b4, 16, 1, b4, 8, 1, 8, p16, 8, 1, p16, 0, p16, 16, None) == -2         # workspace too small
    assert lib.linr_linear_wgrad_wide(None, 16, 1, b4, 8, 1, 8, None, 8, 1, p16, 0, p16, 4096, None) == -1    # no inputs (gW = NULL alone: partials only)
    assert lib.linr_spconv_wgrad_wide_blocks(16, 1) == 256 and lib.linr_spconv_wgrad_wide_blocks(24, 1) == 512 and lib.linr_spconv_wgrad_wide_blocks(16, 0) == 512
    assert lib.linr_linear_wgrad_wide_blocks(0) == 0 and lib.linr_linear_wgrad_wide_blocks(1000) == 4 and lib.linr_linear_wgrad_wide_blocks(10 ** 7) == 512
    from linr_pcgc_amd._lib import LinrWideReduce
    items = (LinrWideReduce * 2)(LinrWideReduce(0, 256, 16, 16, 0, 0, p16, p16, p16), LinrWideReduce(1, 4, 16, 8, 8, 1, p16, p16, None))
    assert lib.linr_wide_reduce_many(items, 0, None) == 0 and lib.linr_wide_reduce_many(None, 2, None) == -1
    items[1].kind = 2
    assert lib.linr_wide_reduce_many(items, 2, None) == -1                                                       # unknown kind
    items[1].kind, items[0].cout = 1, 12
    assert lib.linr_wide_reduce_many(items, 2, None) == -1                                                       # a convolution's cout is a multiple of 8
    items[0].cout, items[0].gW = 16, None
    assert lib.linr_wide_reduce_many(items, 2, None) == -1                                                       # no destination
    assert lib.linr_linear_wgrad_wide(b4, 32, 0, b4, 8, 1, 8, p16, 8, 1, p16, 0, p16, 4096, None) == -1       # a dense side has <= 31 channels
    assert lib.linr_head_wide_fwd(b4, 24, p16, p16, p16, p16, None, 1, 8, p16, None, None, 0, None) == -1     # C is 16 or 32
    assert lib.linr_head_wide_fwd(b4, 16, p16, p16, p16, p16, None, 1, 8, p16, p16, p16, 64, None) == -1      # bits without a target
    assert lib.linr_head_wide_fwd(b4, 16, p16, p16, p16, p16, p16, 8, 8, p16, p16, p16, 4, None) == -2        # workspace too small
    assert lib.linr_head_wide_fwd(b4, 16, p16, p16, p16, p16, None, 1, 0, p16, None, None, 0, None) == 0      # empty input is fine
    assert lib.linr_bits_finish(p16, -1, p16, None) == -1 and lib.linr_bits_finish(None, 4, p16, None) == -1 and lib.linr_bits_finish(p16, 0, p16, None) == 0
    assert lib.linr_head_wide_bwd_slab_bytes(16, 8) == 256 * 8 * (24 * 16 + 49) * 4 and lib.linr_head_wide_bwd_slab_bytes(8, 8) == 0
    assert lib.linr_head_wide_bwd(b4, b4, b4, 8, b4, b4, b4, 16, 9, 1.0, b4, 8, p16, 1 << 30, p16, None) == -1      # more than 8 stages
    assert lib.linr_head_wide_bwd(b4, b4, b4, 8, b4, b4, b4, 16, 1, 1.0, b4, 8, p16, 64, p16, None) == -2           # slab too small
    assert lib.linr_sum_many(b4, 9, 8, p16, 0, None) == -1 and lib.linr_sum_many(b4, 2, 6, p16, 0, None) == -1      # count <= 8, n % 4 == 0
    assert lib.linr_sum_many(b4, 2, 8, p16 + 4, 0, None) == -3 and lib.linr_sum_many(b4, 2, 0, p16, 1, None) == 0
    assert lib.linr_sce_param_count(7) == 7 * 8 + 7 * 392 and lib.linr_sce_param_count(0) == -1
    assert lib.linr_sce_bwd_params_slab_bytes(7) == 256 * (7 * 8 + 7 * 392) * 4
    assert lib.linr_sce_bwd_params(p16, None, p16, p16, p16, 1 << 30, p16, None) == -1                           # no frame

def test_param_count_matches_reference_checkpoint(lib, golden_dir):
    g = np.load(os.path.join(golden_dir, 'loot_model_kat.npz'))
    assert lib.linr_param_count(7, 1) == len(g['flat']) == 54712
    assert lib.linr_param_count(0, 1) < 0 and lib.linr_param_count(17, 1) < 0
    # --block_layers (main.py:521): every extra Inception layer of block_in adds 27*8*4+4 + 2*(27*4*4+4) + 8*4+4 + 4*4+4 floats
    per_layer = (27 * 8 * 4 + 4) + 2 * (27 * 4 * 4 + 4) + (8 * 4 + 4) + (4 * 4 + 4)
    assert lib.linr_param_count(7, 2) == 54712 + per_layer and lib.linr_param_count(7, 3) == 54712 + 2 * per_layer
    assert lib.linr_param_count(7, 0) < 0 and lib.linr_param_count(7, 5) < 0


def test_state_dict_contract(golden_dir):
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    g = np.load(os.path.join(golden_dir, 'loot_model_kat.npz'))
    m = LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8,
                         'instage': 1})
    sd = m.state_dict()
    assert list(sd.keys()) == list(g['names'])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g['shapes'])
    new, off = {}, 0
    for n, v in sd.items():
        new[n] = torch.from_numpy(g['flat'][off:off + v.numel()].reshape(v.shape).copy())
        off += v.numel()
    m.load_state_dict(new)
    assert torch.equal(m.flat_parameters(), torch.from_numpy(g['flat']))          # parameters() order == flat order
    assert torch.equal(torch.cat([p.reshape(-1) for p in m.parameters()]), m.flat_parameters())
    with pytest.raises(ValueError, match='hidden_channel_conv=12 is not supported'):          # 8, 16, 32 are (tests/test_gpu_wide.py)
        LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 12, 'block_layers': 1, 'outstage': 8,
                         'instage': 1})
    wide = LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 16, 'block_layers': 1, 'outstage': 8, 'instage': 1})
    assert wide.state_dict()['upsampler.block_in.3.kernel'].shape == (27, 16, 16) and wide.flat_parameters().numel() == 189944
    with pytest.raises(ValueError, match='block_layers'):
        LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 5, 'outstage': 8,
                         'instage': 1})


def test_block_layers_state_dict_and_mismatch_is_rejected(lib, golden_dir):
    """--block_layers 2 / 3 (main.py:521): the extra Inception layers appear under the reference's names
    (upsampler.block_in.2.layers.<l>.*) right after layer 0, parameters() order == the kernels' flat order, and a
    checkpoint of another depth is rejected by load_state_dict instead of being mis-loaded."""
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    mk = lambda bl: LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': bl,
                                     'outstage': 8, 'instage': 1})
    m1, m2, m3 = mk(1), mk(2), mk(3)
    k1, k2 = list(m1.state_dict()), list(m2.state_dict())
    extra = [k for k in k2 if k not in k1]
    assert extra == ['upsampler.block_in.2.layers.1.%s.%s' % (c, t) for c in ('conv0_0', 'conv0_1', 'conv1_0', 'conv1_1', 'conv1_2')
                     for t in ('kernel', 'bias')]
    assert k2.index(extra[0]) == k2.index('upsampler.block_in.2.layers.0.conv1_2.bias') + 1
    assert [k for k in k2 if k not in extra] == k1
    assert m2.flat_parameters().numel() == lib.linr_param_count(7, 2) and m3.flat_parameters().numel() == lib.linr_param_count(7, 3)
    assert torch.equal(torch.cat([p.reshape(-1) for p in m3.parameters()]), m3.flat_parameters())
    with pytest.raises(RuntimeError, match='layers.1'):
        m1.load_state_dict(m2.state_dict())                     # unexpected keys
    with pytest.raises(RuntimeError, match='layers.1'):
        m2.load_state_dict(m1.state_dict())                     # missing keys


def test_init_statistics():
    """ME conv: U(+-1/sqrt(Cin*K)); PointwiseMLP: xavier_uniform(gain sqrt 2), zero bias; Embedding N(0,1)."""
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    torch.manual_seed(0)
    m = LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8,
                         'instage': 1})
    k = m.upsampler.block_in[0].kernel
    bound = 1 / np.sqrt(8 * 27)
    assert float(k.abs().max()) <= bound and float(k.abs().max()) > 0.9 * bound
    assert float(m.scale_mlp[0][0].bias.abs().max()) == 0.0
    w = m.upsampler.inner_mlps[0][0][0].weight
    assert float(w.abs().max()) <= np.sqrt(2) * np.sqrt(6 / (8 + 24)) + 1e-6


def test_no_gpu_means_loud_failure():
    from linr_pcgc_amd import _lib, engine
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.LinrError):
        engine.Frame([], 7, device='cpu')


@pytest.mark.parametrize('n', [0, 1, 2, 33, 100003])
def test_cpp_binary_coder_bit_exact_with_oracle(lib, n):
    from linr_pcgc_amd.module_utils import BinaryArithmeticCoding
    rng = np.random.default_rng(n)
    p = rng.random(n).astype(np.float32)
    s = (rng.random(n) < p).astype(np.int16)
    if n > 8:
        p[:6] = [0.0, 1.0, 1e-9, 1 - 1e-7, 0.5, 0.25]
        s[:4] = [1, 0, 1, 0]
    bac = BinaryArithmeticCoding()
    data = bac.encode(torch.from_numpy(p).reshape(-1, 1), torch.from_numpy(s))
    assert data == oac.encode_binary(p, s)
    assert (bac.decode(torch.from_numpy(p), data).numpy() == s).all()
    assert (oac.decode_binary(p, data) == s).all()


@pytest.mark.parametrize('seed', [0, 1])
def test_cpp_binary_decoder_skewed_streams_and_truncation(lib, seed):
    """Streams of a trained model are skewed (most symbols cost ~0.1 bit: long runs without renormalisation, then many shifts
    and straddle steps at once), and a truncated stream reads as zeros past its end: the product decoder follows the oracle's
    (one bit at a time, like torchac) in both."""
    from linr_pcgc_amd.module_utils import BinaryArithmeticCoding
    rng = np.random.default_rng(seed)
    n = 60001
    z = rng.normal(-5.5, 2.5, n)
    p = (1 / (1 + np.exp(-z))).astype(np.float32)
    p = np.where(rng.random(n) < 0.5, 1 - p, p).astype(np.float32)
    p[::97] = 0.5 + (rng.random(len(p[::97])).astype(np.float32) - 0.5) * 1e-4          # straddles of the middle
    s = (rng.random(n) < p).astype(np.int16)
    bac = BinaryArithmeticCoding()
    data = bac.encode(torch.from_numpy(p).reshape(-1, 1), torch.from_numpy(s))
    assert data == oac.encode_binary(p, s)
    assert (bac.decode(torch.from_numpy(p), data).numpy() == s).all()
    for cut in (len(data) - 1, len(data) // 2, 3, 0):
        short = data[:cut]
        assert (bac.decode(torch.from_numpy(p), short).numpy() == oac.decode_binary(p, short)).all()


def test_cpp_generic_coder_model_stream(lib, golden_dir):
    from linr_pcgc_amd import model_codec
    g = np.load(os.path.join(golden_dir, 'loot_model_kat.npz'))
    out = model_codec.compress_params(torch.from_numpy(g['flat']), 8)
    ref = omc.encode_model(g['flat'], 8)
    assert out['final_bytes'] == ref['bytes'] and out['enc_mode'] == 2
    assert (out['mu'], out['b']) == (float(g['mu']), float(g['b']))
    assert out['min_param'] == float(g['min_param']) and out['max_param'] == float(g['max_param'])
    assert len(out['final_bytes']) == 35319                     # pinned with its derivation in tests/test_oracle_golden.py::test_model_stream_known_answer
    rec = model_codec.decompress_params(out, len(g['flat']))
    assert torch.equal(rec, ref['recon'])


def test_batch_encoder_threads(lib):
    from linr_pcgc_amd.model_core import encode_streams
    rng = np.random.default_rng(9)
    ps = [rng.random(n).astype(np.float32) for n in (0, 5, 1000, 40000, 7)]
    ss = [(rng.random(len(p)) < p).astype(np.uint8) for p in ps]
    got = encode_streams(ps, ss, n_threads=4)
    assert got == [oac.encode_binary(p, s.astype(np.int16)) for p, s in zip(ps, ss)]


def test_pack_bitstream_golden(golden_dir):
    from linr_pcgc_amd.function_utils import pack_bitstream, unpack_bitstream
    g = np.load(os.path.join(golden_dir, 'pack_bitstream.npz'))
    payload, lens = g['payload'].tobytes(), g['lens']
    streams, pos = [], 0
    for n in lens:
        streams.append(payload[pos:pos + int(n)])
        pos += int(n)
    packed = pack_bitstream(streams)
    assert packed == g['packed'].tobytes()
    assert [bytes(b) for b in unpack_bitstream(packed)] == streams


@pytest.mark.parametrize('name', ['octree_random64.npz', 'octree_shell128.npz'])
def test_product_octree_prep_golden(golden_dir, name):
    from linr_pcgc_amd.module_utils import octree_level_obj, prepare_frame
    g = np.load(os.path.join(golden_dir, name))
    fr = prepare_frame(g['points'], None, int(g['min_point_num']))
    assert fr['scale_num'] == int(g['scale_num'])
    assert fr['coord_data_min'] == g['coord_data_min'].tolist()
    assert (fr['ori'].numpy() == g['ori']).all()
    for s, info in enumerate(fr['all_input_info']):
        assert (info['coord'].numpy() == g['s%d_coord' % s]).all()
        assert (info['occ'].numpy() == g['s%d_occ' % s]).all()
        assert (info['offset_tensor'].numpy() == g['s%d_offset' % s]).all()
        assert (octree_level_obj.upper_layer(info['coord'], info['occ']).numpy() == g['s%d_upper' % s]).all()


def test_synthetic_configs():
    from linr_pcgc_amd import synthetic
    from linr_pcgc_amd.module_utils import prepare_frame
    pts = synthetic.sphere_shell(8, 100)
    assert len(pts) == 125810
    fr = prepare_frame(pts)
    assert fr['all_input_info'][0]['coord'].shape[0] == 39305 and fr['scale_num'] == 6
    key = pts[:, 0].astype(np.int64) << 40 | pts[:, 1].astype(np.int64) << 20 | pts[:, 2]
    assert (np.diff(key) > 0).all()


def test_rough_figure_generator_is_a_deterministic_non_spherical_surface():
    """synthetic.rough_figure (the non-spherical stress workload beside the sphere stand-ins; the reference's data are
    human figures, datautils/custom_dataset.py:259-355): deterministic, x-major sorted and unique like every stand-in, in range, a
    figure and not a ball (anisotropic extent, radii spread widely about the centroid where a sphere shell's do not), integer motion
    from frame to frame, and a kernel map whose taps-per-row lie in the range of a surface (between a curve's 3 and a solid's 27)."""
    from linr_pcgc_amd import synthetic
    from oracle import octree
    a, a2, b = synthetic.rough_figure(7, 0), synthetic.rough_figure(7, 0), synthetic.rough_figure(7, 1)
    assert a.dtype == np.int32 and a.shape == (16412, 3) and np.array_equal(a, a2)
    assert a.min() >= 0 and a.max() < 128
    key = a[:, 0].astype(np.int64) << 40 | a[:, 1].astype(np.int64) << 20 | a[:, 2]
    assert (np.diff(key) > 0).all()
    ext = (a.max(0) - a.min(0)).astype(float)
    assert ext.max() >= 2.5 * ext.min()                                      # taller than deep
    c = a.astype(float)
    r = np.linalg.norm(c - c.mean(0), axis=1)
    assert r.std() / r.mean() > 0.3                                          # a 1-voxel sphere shell: < 0.01
    s = synthetic.sphere_shell(7, 50).astype(float)
    rs = np.linalg.norm(s - s.mean(0), axis=1)
    assert rs.std() / rs.mean() < 0.01
    assert b.shape != a.shape or not np.array_equal(a, b)                    # the figure moves
    assert abs(len(b) - len(a)) < 0.02 * len(a)
    taps = (octree.neighbour_table(a.astype(np.int64)) >= 0).sum(1).mean()
    assert 9.0 < taps < 18.0, taps


def test_committed_bench_line_keeps_the_driver_contract():
    """The line bench.py printed for the driver's command at the end of the round (profiles/r06_final_bench_driver_cmd.json): every key
    of the bench contract, BASELINE.json's metric and unit, the roofline and cpu_baseline objects with their fields, a lossless run."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r0*_final_bench_driver_cmd.json')))
    assert files
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 's/frame' and d['higher_is_better'] is False and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['n_gpus'] == 1 and 'workload' in d['config'] and 'model' not in d['config']
    r, c = d['roofline'], d['cpu_baseline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is None or r['traffic'] > 0
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit'] and c['sample']
    assert d['lossless_decode_frames0to3'] is True and d['sequence']['lossless'] is True
    assert 0 < d['value'] < 1 and abs(d['value'] - (d['components_s_per_frame']['overfit'] + d['components_s_per_frame']['codec_modelcomp_fwd_ac_write'])) < 2e-5


def test_flat_adam_state_dict_roundtrip():
    from linr_pcgc_amd.model_core import FlatAdam, LINR_PCGC_Model
    m = LINR_PCGC_Model({'scale_num': 6, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8,
                         'instage': 1})
    opt = FlatAdam(m)
    opt.exp_avg.normal_()
    opt.exp_avg_sq.uniform_()
    opt.t, opt.lr = 17, 0.00731
    opt.t_scale[:] = [17, 17, 17, 17, 9, 0]                     # scale 4 was absent from 8 frames, scale 5 never seen
    sd = opt.state_dict()
    names = [n for n, _ in m.named_parameters()]
    assert all((i in sd['state']) == (not names[i].startswith('scale_mlp.5.')) for i in range(len(names)))
    assert float(sd['state'][names.index('scale_mlp.4.0.weight')]['step']) == 9.0
    ref = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-4)
    ref.load_state_dict(sd)                                     # torch accepts our format
    opt2 = FlatAdam(m)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.t == 17 and abs(opt2.lr - 0.00731) < 1e-12 and opt2.t_scale.tolist() == [17, 17, 17, 17, 9, 0]
    keep = torch.cat([torch.full((p.numel(),), 0.0 if n.startswith('scale_mlp.5.') else 1.0) for n, p in m.named_parameters()])
    assert torch.equal(opt2.exp_avg, opt.exp_avg * keep) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq * keep)
    for i in range(1, 100):
        opt2.scheduler_step()
    assert abs(opt2.lr - 0.00731 * 0.992 ** 3) < 1e-12          # StepLR(32, 0.992) stepped per frame


@pytest.mark.parametrize('binary', [True, False])
def test_ply_reader(tmp_path, binary):
    """read_ply_o3d replacement (custom_dataset.py:9-14): x, y, z from ascii / binary PLY with extra vertex properties."""
    from linr_pcgc_amd import ply
    rng = np.random.default_rng(4)
    xyz = rng.integers(0, 1024, size=(257, 3))
    path = str(tmp_path / 'a.ply')
    ply.write_ply_xyz(path, xyz, binary=binary)
    assert np.array_equal(ply.read_points(path), xyz)
    # a vertex element with colours and a face element behind it, big endian
    path2 = str(tmp_path / 'b.ply')
    rec = np.zeros(5, dtype=[('x', '>f8'), ('red', 'u1'), ('y', '>f8'), ('z', '>f8'), ('green', 'u1')])
    rec['x'], rec['y'], rec['z'] = [1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [11, 12, 13, 14, 15]
    with open(path2, 'wb') as f:
        f.write(b'ply\nformat binary_big_endian 1.0\nelement vertex 5\nproperty double x\nproperty uchar red\n'
                b'property double y\nproperty double z\nproperty uchar green\nelement face 0\n'
                b'property list uchar int vertex_indices\nend_header\n')
        f.write(rec.tobytes())
    got = ply.read_points(path2)
    assert got.tolist() == [[1, 6, 11], [2, 7, 12], [3, 8, 13], [4, 9, 14], [5, 10, 15]]
    np.save(str(tmp_path / 'c.npy'), xyz)
    assert np.array_equal(ply.read_points(str(tmp_path / 'c.npy')), xyz)
    with pytest.raises(ValueError):
        open(str(tmp_path / 'd.ply'), 'wb').write(b'plx\n')
        ply.read_points(str(tmp_path / 'd.ply'))


def test_decoder_refuses_streams_of_another_arithmetic_version():
    """The decoder must reproduce the encoder's probabilities bit for bit, so the fp32 evaluation order of the network is part of
    the stream format (codec.ARITH_VERSION, written to side_info.json): a stream of another version is refused before anything
    touches the GPU; streams without the tag are version 1."""
    from linr_pcgc_amd import codec
    from linr_pcgc_amd._lib import LinrError
    assert codec.ARITH_VERSION >= 2
    for side in ({'arith_version': codec.ARITH_VERSION - 1}, {}):
        with pytest.raises(LinrError, match='arithmetic version'):
            codec.decode_gop(None, {'side_info': side, 'model_bin': b'', 'low_enc_bytes': b'', 'frames': []})


def test_ascii_ply_parser_against_numpy(tmp_path):
    """linr_ply_parse_ascii (include/linr_hip.h; read_ply_o3d of custom_dataset.py:9-14 for the ASCII files the data sets ship):
    the same integers as numpy's text reader + rint on a loot-style body (x y z as floats, colours behind), on decimals, signs,
    exponents, ties (round half to even), CRLF line ends, blank lines; malformed bodies raise with the line number."""
    import ctypes
    from linr_pcgc_amd import _lib, ply
    rng = np.random.default_rng(11)
    n = 5000
    xyz = rng.integers(0, 1024, size=(n, 3))
    cols = np.concatenate([rng.integers(0, 256, size=(n, 2)), xyz[:, 2:3], xyz[:, 0:1], rng.integers(0, 256, size=(n, 1)), xyz[:, 1:2]], axis=1)
    head = ('ply\nformat ascii 1.0\ncomment generated\nelement vertex %d\nproperty uchar red\nproperty uchar green\nproperty float z\n'
            'property float x\nproperty uchar blue\nproperty float y\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n' % n)
    path = str(tmp_path / 'loot_like.ply')
    with open(path, 'w') as f:
        f.write(head)
        for i, row in enumerate(cols):
            if i % 3 == 0:
                f.write('%d %d %.6f %.1f %d %d.000\r\n' % tuple(row))          # float spellings, CRLF
            elif i % 3 == 1:
                f.write('  %d\t%d %d %d  %d %d \n\n' % tuple(row))              # blanks, tabs, an empty line behind
            else:
                f.write('%d %d %de0 +%d %d %.3e\n' % tuple(row))               # exponent forms -> the strtod path
    assert np.array_equal(ply.read_points(path), xyz)
    assert all(np.array_equal(a, xyz) for a in ply.read_many([path] * 3, workers=2))
    # rounding and signs, straight through the C entry
    text = b'0.5 1.5 2.5\n-0.5 -1.5 -2.5\n2.4999 -7.50001 1e3\n'
    out = np.empty((3, 3), dtype=np.int64)
    done = ctypes.c_int64(0)
    L = _lib.lib()
    assert L.linr_ply_parse_ascii(text, len(text), 3, 3, 0, 1, 2, out.ctypes.data, ctypes.byref(done)) == 0 and done.value == 3
    want = np.rint(np.loadtxt(text.decode().splitlines(), ndmin=2)).astype(np.int64)
    assert np.array_equal(out, want) and out.tolist() == [[0, 2, 2], [0, -2, -2], [2, -8, 1000]]
    # malformed: a short line, a word, a non-finite coordinate, fewer vertices than announced, bad column arguments
    for bad, at in ((b'1 2 3\n4 5\n6 7 8\n', 1), (b'1 2 3\n4 x 6\n', 1), (b'1 2 nan\n', 0), (b'1 2 3\n', 1), (b'1 2 3 4\n5 6 7\n', 0)):
        rc = L.linr_ply_parse_ascii(bad, len(bad), 2 if at else 1, 3, 0, 1, 2, out.ctypes.data, ctypes.byref(done))
        assert rc == -1 and done.value == at, (bad, rc, done.value)
    assert L.linr_ply_parse_ascii(text, len(text), 3, 3, 0, 1, 3, out.ctypes.data, None) == -1
    assert L.linr_ply_parse_ascii(text, len(text), 3, 2, 0, 1, 1, out.ctypes.data, None) == -1
    with open(str(tmp_path / 'short.ply'), 'w') as f:
        f.write('ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nend_header\n1 2 3\n4 5 6\n')
    with pytest.raises(ValueError, match='line 3'):
        ply.read_points(str(tmp_path / 'short.ply'))


def test_model_size_estimate_agrees_with_the_real_model_codec():
    """Model_Estimate.estibits (model_size_est.py:99-179): main.py:290-295 runs it next to compress_test before training and
    asserts the two reconstructions equal; the estimated size must also sit within a per cent of the coded one in Laplace mode."""
    from linr_pcgc_amd.model_codec import Model_Estimate, esti_model_size
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    cfg = {'scale_num': 3, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8, 'instage': 1}
    torch.manual_seed(3)
    model = LINR_PCGC_Model(cfg)
    est = Model_Estimate().estibits(model, LINR_PCGC_Model(cfg), 8)
    real = Model_Estimate().compress_test(model, LINR_PCGC_Model(cfg), 8)
    assert int((est['recon_ret'] != real['recon_ret']).sum()) == 0
    assert torch.equal(est['new_model'].flat_parameters(), real['new_model'].flat_parameters())
    assert est['enc_mode'] == real['enc_mode'] and float(est['mu']) == real['mu'] and float(est['b']) == real['b']
    if est['enc_mode'] == 2:
        assert abs(est['bit_real'] - real['bit_real']) <= 0.01 * real['bit_real']
    assert esti_model_size(model) == 32 * model.flat_parameters().numel()
    assert sorted(est) == ['b', 'bit_real', 'bpp_real', 'dec_time', 'enc_mode', 'enc_time', 'final_bytes', 'laplace_bpp', 'max_param',
                           'min_param', 'mu', 'new_model', 'recon_ret', 'zlib_bpp']


def test_dataset_classes_of_the_drivers(tmp_path):
    """datautils/custom_dataset.py's MyDataset / Read_Data / MytestDataset (main.py:73-78,110, encoder.py:47, decoder.py:118-131) on
    this package's octree preparation, against the oracle's restatement: same per-scale inputs, scale_num fixed by frame 0, frames
    cached in RAM, the decoder's sorted voxel list."""
    from linr_pcgc_amd import custom_dataset as cd
    rng = np.random.default_rng(21)
    ori = tmp_path / 'ori'
    ori.mkdir()
    clouds = []
    for t in range(3):
        c = rng.integers(5, 69, size=(6000, 3))
        c = np.concatenate([c, c[:50]], axis=0)[rng.permutation(6050)]          # duplicates, no order
        clouds.append(c)
        if t == 2:
            cd.write_ply_ascii(str(ori / ('f%03d.ply' % t)), c)
        np.save(str(ori / ('f%03d.npy' % t)), c)
    (ori / 'subdir.npy').mkdir()
    ds = cd.MyDataset(str(ori), str(tmp_path / 'handle'), None, 'npy', stage=8, derive_ori=True)
    ds.set_prefix_data({'offsets_ini': [[0, 0, 0], [-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]], 'min_point_num': 64})
    assert len(ds.all_files_path) == 3 and os.path.isdir(str(tmp_path / 'handle')) and ds.scale_num is None
    first = ds[0]
    assert ds.scale_num == len(first['all_input_info']) and ds[0] is first          # RAM cache: the same object
    for t in range(3):
        got, want = ds[t], ooct.prepare_frame(clouds[t], ds.scale_num, 64)
        assert got['point_num'] == want['point_num'] and got['coord_data_min'] == [int(v) for v in want['coord_data_min']]
        assert np.array_equal(got['ori'].cpu().numpy(), want['ori'])
        assert len(got['all_input_info']) == len(want['scales'])
        for a, b in zip(got['all_input_info'], want['scales']):
            assert a['scale_idx'] == b['scale_idx']
            assert np.array_equal(a['xyzqsc_t'].get_coord().cpu().numpy(), b['coord'])
            assert np.array_equal(torch.cat(a['occ_lst'], dim=1).cpu().numpy(), b['occ']) and len(a['occ_lst']) == 8
            assert np.array_equal(a['xyzqsc_t'].get_offset_tensor().cpu().numpy(), b['offset_tensor'])
        low = got['all_input_info'][-1]['xyzqsc_t'].get_coord()
        bd = int(np.ceil(np.log2(int(low.max()) + 1)))
        assert got['xyzQ_low_bits'] == min(len(low), 8 ** bd - len(low)) * bd * 3
    win = cd.Read_Data_with_cache(ds, [1, 2])
    assert len(win) == 2 and win[0] is ds[1] and win[1] is ds[2]
    assert len(cd.MyDataset(str(ori), None, 2, 'npy', stage=4)[0]['all_input_info'][0]['occ_lst']) == 4
    test = cd.MytestDataset(str(ori), ori_type='ply')
    assert len(test) == 1
    srt = test[0].cpu().numpy()
    key = lambda a: (a[:, 0].astype(np.int64) << 42) | (a[:, 1].astype(np.int64) << 21) | a[:, 2].astype(np.int64)
    assert srt.shape == (6050, 3) and np.all(np.diff(key(srt)) >= 0) and np.array_equal(np.unique(srt, axis=0), np.unique(clouds[2], axis=0))
    assert np.array_equal(cd.read_ply_o3d(str(ori / 'f002.ply')), clouds[2])
    with pytest.raises(ValueError):
        cd.MyDataset(str(tmp_path / 'handle'), None, None, 'npy')                     # no frame files there
    with pytest.raises(ValueError):
        ds.set_prefix_data({'offsets_ini': [[0, 0, 0], [2, 0, 0]]})


@pytest.mark.parametrize('bitdepth', [4, 6, 8, 10, 16])
def test_model_codec_round_trip_at_other_bit_depths(bitdepth):
    """--model_bitdepth (main.py:522): compress_test's own consistency check (coded -> decoded reconstruction == the encoder's) at
    depths below and above the default 8; above 8 the codes travel as uint16 (raw / zlib modes - the reference's decoder reads them
    as uint8 there and fails), and the quantisation error shrinks with the depth."""
    from linr_pcgc_amd.model_codec import Model_Estimate
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    cfg = {'scale_num': 3, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8, 'instage': 1}
    torch.manual_seed(5)
    model = LINR_PCGC_Model(cfg)
    out = Model_Estimate().compress_test(model, LINR_PCGC_Model(cfg), bitdepth)          # asserts recon == decoded recon itself
    flat = model.flat_parameters()
    step = float(flat.max() - flat.min()) / (2 ** bitdepth - 1)
    assert float((out['new_model'].flat_parameters() - flat).abs().max()) <= 0.5 * step + 1e-6          # + fp32 rounding of the affine map
    assert out['enc_mode'] in (0, 1, 2) and (bitdepth <= 8 or out['enc_mode'] in (0, 1))
    assert out['bit_real'] <= bitdepth * flat.numel() + 2 + 64 + 2 * bitdepth


def test_prepare_frame_refuses_what_it_cannot_represent():
    from linr_pcgc_amd.module_utils import prepare_frame
    with pytest.raises(ValueError, match='no points'):
        prepare_frame(np.zeros((0, 3), np.int64))
    with pytest.raises(ValueError, match='20-bit'):
        prepare_frame(np.array([[0, 0, 0], [1 << 20, 5, 5]]))
    fr = prepare_frame(np.array([[0, 0, 0], [(1 << 20) - 1, 5, 5]]), device='cpu')          # the widest cloud that fits
    assert fr['point_num'] == 2


def test_host_code_under_sanitizers(tmp_path):
    """The host-side C++ (range coder, ASCII PLY parser) under AddressSanitizer + UBSan: tools/host_fuzz.cpp round-trips random
    streams through buffers of the exact size, decodes truncated / random streams, refuses short output buffers, and parses valid,
    mutated and truncated PLY bodies.  Any out-of-bounds access or undefined operation aborts the program."""
    import shutil
    import subprocess
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'host_fuzz')
    build = subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                            '-fno-omit-frame-pointer', '-include', 'algorithm', '-o', exe, os.path.join(root, 'tools', 'host_fuzz.cpp'),
                            os.path.join(root, 'linr_pcgc_amd', 'csrc', 'ac.cpp'), os.path.join(root, 'linr_pcgc_amd', 'csrc', 'ply.cpp'),
                            '-lpthread'], capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and 'asan' in (build.stderr or '').lower():
        pytest.skip('sanitizer runtime not installed')
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, '400'], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and 'fuzz ok: 400 iterations' in run.stdout, (run.stdout[-500:], run.stderr[-3000:])


def test_reference_checkpoint_optimizer_state_loads(golden_dir):
    """GOPs >= 1 load GOP 0's `optimizer_state_dict` besides the weights (main.py:241-248).  The state inside the checkpoint the
    reference ships - written by torch 1.13.1's Adam at epoch 70 (tests/golden/loot_optimizer_state.npz) - goes into FlatAdam
    (learning rate as decayed, 7,223 steps on every tensor, both moments bit for bit) and comes back out in torch's format."""
    from linr_pcgc_amd.model_core import FlatAdam, LINR_PCGC_Model
    g = np.load(os.path.join(golden_dir, 'loot_optimizer_state.npz'))
    m = LINR_PCGC_Model({'scale_num': 7, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8, 'instage': 1})
    sizes = [int(v) for v in g['sizes']]
    assert sizes == [p.numel() for p in m.parameters()]          # 189 tensors in the reference's registration order
    ea, es = torch.from_numpy(g['exp_avg']), torch.from_numpy(g['exp_avg_sq'])
    state, off = {}, 0
    for i, (p, n) in enumerate(zip(m.parameters(), sizes)):
        state[i] = {'step': torch.tensor(float(g['step'][i])), 'exp_avg': ea[off:off + n].view(p.shape).clone(),
                    'exp_avg_sq': es[off:off + n].view(p.shape).clone()}
        off += n
    sd = {'state': state, 'param_groups': [{'lr': float(g['lr']), 'betas': (float(g['beta1']), float(g['beta2'])), 'eps': float(g['eps']),
                                            'weight_decay': float(g['weight_decay']), 'amsgrad': bool(g['amsgrad']), 'maximize': False,
                                            'foreach': None, 'capturable': False, 'initial_lr': float(g['initial_lr']),
                                            'params': list(range(len(sizes)))}]}
    opt = FlatAdam(m)
    opt.load_state_dict(sd)
    assert opt.lr == float(g['lr']) and 0 < opt.lr < 0.01 and opt.t == 7223 and opt.t_scale.tolist() == [7223] * 7
    assert opt.betas == (0.9, 0.999) and opt.eps == 1e-8 and opt.weight_decay == 1e-4
    assert torch.equal(opt.exp_avg.cpu(), ea) and torch.equal(opt.exp_avg_sq.cpu(), es)
    back = opt.state_dict()
    assert all(torch.equal(back['state'][i]['exp_avg'].cpu(), state[i]['exp_avg']) and float(back['state'][i]['step']) == 7223.0 for i in range(len(sizes)))
    ref = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-4)
    ref.load_state_dict(back)                                    # and torch takes it back
    assert ref.param_groups[0]['lr'] == float(g['lr'])


def test_run_train_precision_flag():
    """run.py: --train-precision selects the overfit's arithmetic; without it the overfit follows --precision (BASELINE config[4]:
    '--precision bf16' = bf16 SparseConv for the overfit and the codec), except where the bf16 training executor does not exist
    (hidden_channel_conv 16 / 32, block_layers > 1)."""
    from linr_pcgc_amd import run
    assert run.train_precision(run.parse([])) == 'f32'
    assert run.train_precision(run.parse(['--precision', 'bf16'])) == 'bf16'
    assert run.train_precision(run.parse(['--precision', 'bf16', '--train-precision', 'f32'])) == 'f32'
    assert run.train_precision(run.parse(['--train-precision', 'bf16'])) == 'bf16'
    assert run.train_precision(run.parse(['--precision', 'bf16', '--block_layers', '2'])) == 'f32'
    assert run.train_precision(run.parse(['--precision', 'bf16', '--hidden-channel-conv', '16'])) == 'f32'
