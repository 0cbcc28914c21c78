"""GPU parity of the single operators through the C-ABI against the CPU oracle: kernel map, octree occupancy, conv3 forward /
backward (gather and compressed-map MFMA forms), pointwise layers, BCE bits, Adam.  Tolerances: tests/gpu_common.py.
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


# ---- kernel map -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['shell', 'random', 'line', 'single', 'empty'])
def test_kmap_bit_exact(pkg, shell, case):
    from linr_pcgc_amd import ops
    rng = np.random.default_rng(5)
    if case == 'shell':
        c = shell['scales'][0]['coord']
    elif case == 'random':
        c = ooct.unique_sorted(rng.integers(0, 40, size=(30000, 3)))
    elif case == 'line':
        c = ooct.unique_sorted(np.stack([np.arange(5000) % 1000, np.zeros(5000, int), np.arange(5000) // 1000], 1))
    elif case == 'single':
        c = np.array([[(1 << 20) - 1, 0, 7]], dtype=np.int32)
    else:
        c = np.zeros((0, 3), dtype=np.int32)
    nbr = ops.kmap_build(torch.from_numpy(c).to(_dev()))
    ref = ooct.neighbour_table(c) if len(c) else np.zeros((0, 27), np.int32)
    assert nbr.shape == (27, len(c))
    assert (nbr.t().cpu().numpy() == ref).all()


def test_octree_occupancy_kernel_matches_golden(pkg, golden_dir):
    """linr_octree_occupancy against the fixtures the reference's own octree_level produced (bit-exact)."""
    from linr_pcgc_amd import ops
    dev = _dev()
    for name in ('octree_random64.npz', 'octree_shell128.npz'):
        g = np.load(os.path.join(golden_dir, name))
        child = torch.from_numpy(g['ori'].astype(np.int32)).to(dev)
        for s in range(int(g['scale_num'])):
            parent = torch.from_numpy(g['s%d_coord' % s].astype(np.int32)).to(dev)
            occ = ops.octree_occupancy(child.contiguous(), parent.contiguous())
            assert np.array_equal(occ.cpu().numpy(), g['s%d_occ' % s].astype(np.float32)), (name, s)
            child = parent
    empty = ops.octree_occupancy(torch.zeros((0, 3), dtype=torch.int32, device=dev), torch.zeros((0, 3), dtype=torch.int32, device=dev))
    assert empty.shape == (0, 8)


def test_offset_features_from_kernel_map(pkg, shell):
    """linr_kmap_offset_feat == qscTensor.set_offset_tensor (the reference's 7 coordinate searches), bit for bit."""
    from linr_pcgc_amd import engine
    dev = _dev()
    scales = [{'coord': s['coord'], 'offset_tensor': None, 'scale_idx': s['scale_idx']} for s in shell['scales']]
    f = engine.Frame(scales, len(scales), dev, with_arena=False)
    ref = np.concatenate([s['offset_tensor'] for s in shell['scales']], axis=0)
    assert np.array_equal(f.offset_feat.cpu().numpy(), ref.astype(np.float32))


def test_kmap_rejects_unsorted(pkg):
    from linr_pcgc_amd import ops
    c = torch.tensor([[1, 0, 0], [0, 0, 0]], dtype=torch.int32, device=_dev())
    with pytest.raises(ValueError):
        ops.kmap_build(c)
    with pytest.raises(ValueError):
        ops.kmap_build(torch.tensor([[0, 0, 0], [0, 0, 0]], dtype=torch.int32, device=_dev()))
    with pytest.raises(ValueError):                                      # coordinate range is [0, 2^20)
        ops.kmap_build(torch.tensor([[0, 0, 0], [0, 0, 1 << 20]], dtype=torch.int32, device=_dev()))
    with pytest.raises(ValueError):
        ops.kmap_build(torch.tensor([[-1, 0, 0], [0, 0, 0]], dtype=torch.int32, device=_dev()))
    top = (1 << 20) - 1                                                  # the largest legal coordinate still maps
    nbr = ops.kmap_build(torch.tensor([[top, top, top - 1], [top, top, top]], dtype=torch.int32, device=_dev()))
    assert nbr[13].tolist() == [0, 1] and nbr[22, 0].item() == 1 and nbr[4, 1].item() == 0


# ---- sparse convolution -------------------------------------------------------------------------------------------------
CONV_SHAPES = [(8, 8), (8, 4), (4, 4)] + [(k, 8) for k in range(1, 8)]


@pytest.mark.parametrize('cin,cout', CONV_SHAPES)
@pytest.mark.parametrize('pad', [False, True])
def test_spconv_fwd_bwd(pkg, shell, cin, cout, pad):
    from linr_pcgc_amd import ops
    dev = _dev()
    sc = shell['scales'][0]
    n = len(sc['coord'])
    g = torch.Generator().manual_seed(100 * cin + cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(27, cin, cout, generator=g) * 0.2
    b = torch.randn(1, cout, generator=g)
    res = torch.randn(n, cout, generator=g)
    go = torch.randn(n, cout, generator=g)
    nbr_o = torch.from_numpy(sc['nbr']).long()
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.relu(onet.conv3(xo, nbr_o, wo, bo) + res)
    ref_lin = onet.conv3(xo, nbr_o, wo, bo)
    ref_lin.backward(go)

    nbr = ops.kmap_build(torch.from_numpy(sc['coord']).to(dev))

    def padded(t):                       # [1+n, ld] with a zero row in front (LINR_PAD_ROW contract)
        buf = torch.zeros((n + 1, t.shape[1]), device=dev)
        buf[1:] = t.to(dev)
        return buf[1:]
    xd = padded(x) if pad else x.to(dev)
    god = padded(go) if pad else go.to(dev)
    out = ops.spconv_fwd(xd, nbr, w.to(dev), b.to(dev), res=res.to(dev), relu=True, pad_row=pad)
    _close(out, ref, 1e-4, 1e-4, 'fwd')
    out2 = ops.spconv_fwd(xd, nbr, w.to(dev), b.to(dev), pad_row=pad)
    assert torch.equal(out2, ops.spconv_fwd(xd, nbr, w.to(dev), b.to(dev), pad_row=pad)), 'fwd must be bit-reproducible'
    gin = ops.spconv_bwd_data(god, nbr, w.to(dev), pad_row=pad)
    _close(gin, xo.grad, 1e-4, 1e-4, 'bwd_data')
    gw, gb = ops.spconv_bwd_weight(xd, god, nbr, cin, cout, pad_row=pad)
    scale = float(wo.grad.abs().max())
    _close(gw, wo.grad, 0, 1e-4 * scale + 1e-6, 'bwd_weight')
    _close(gb, bo.grad, 0, 1e-4 * float(bo.grad.abs().max()) + 1e-6, 'bwd_bias')
    gw2, _ = ops.spconv_bwd_weight(xd, god, nbr, cin, cout, pad_row=pad)
    assert torch.equal(gw, gw2), 'bwd_weight must be bit-reproducible (two-pass, no atomics)'


def test_spconv_pad_equals_branch_bitwise(pkg, shell):
    """fmaf(0, w, acc) == acc: the zero-row variant and the branch variant must agree bit for bit."""
    from linr_pcgc_amd import ops
    dev = _dev()
    sc = shell['scales'][1]
    n = len(sc['coord'])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, 8, generator=g)
    w, b = torch.randn(27, 8, 8, generator=g).to(dev), torch.randn(1, 8, generator=g).to(dev)
    nbr = ops.kmap_build(torch.from_numpy(sc['coord']).to(dev))
    buf = torch.zeros((n + 1, 8), device=dev)
    buf[1:] = x.to(dev)
    assert torch.equal(ops.spconv_fwd(buf[1:], nbr, w, b, pad_row=True), ops.spconv_fwd(x.to(dev), nbr, w, b))


def test_spconv_channel_slices(pkg, shell):
    """ME.cat / merge_two_frames are pointer offsets here: read a 4-channel slice, write into a slice."""
    from linr_pcgc_amd import ops
    dev = _dev()
    sc = shell['scales'][0]
    n = len(sc['coord'])
    g = torch.Generator().manual_seed(3)
    x8 = torch.randn(n, 8, generator=g)
    w, b = torch.randn(27, 4, 4, generator=g) * 0.3, torch.randn(1, 4, generator=g)
    nbr = ops.kmap_build(torch.from_numpy(sc['coord']).to(dev))
    ref = onet.conv3(x8[:, 4:8], torch.from_numpy(sc['nbr']).long(), w, b)
    out8 = torch.full((n, 8), 7.0, device=dev)
    ops.spconv_fwd(x8.to(dev)[:, 4:8], nbr, w.to(dev), b.to(dev), out=out8[:, 0:4])
    _close(out8[:, 0:4], ref, 1e-4, 1e-4, 'slice fwd')
    assert bool((out8[:, 4:8] == 7.0).all())


# ---- pointwise ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,layout', [(15, 16, 'torch'), (16, 8, 'torch'), (8, 24, 'torch'), (24, 1, 'torch'),
                                             (8, 4, 'me'), (4, 4, 'me')])
def test_linear_fwd_bwd(pkg, cin, cout, layout):
    from linr_pcgc_amd import ops
    dev = _dev()
    n = 10007
    g = torch.Generator().manual_seed(cin * 31 + cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn((cin, cout) if layout == 'me' else (cout, cin), generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    go = torch.randn(n, cout, generator=g)
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = (xo @ wo if layout == 'me' else xo @ wo.t()) + bo
    ref.backward(go)
    out = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), cin, cout, layout)
    _close(out, ref, 1e-4, 1e-4, 'fwd')
    gin = ops.linear_bwd_data(go.to(dev), w.to(dev), cin, cout, layout)
    _close(gin, xo.grad, 1e-4, 1e-4, 'bwd_data')
    gw, gb = ops.linear_bwd_weight(x.to(dev), go.to(dev), cin, cout, layout)
    _close(gw, wo.grad, 0, 1e-4 * float(wo.grad.abs().max()) + 1e-6, 'bwd_weight')
    _close(gb, bo.grad, 0, 1e-4 * float(bo.grad.abs().max()) + 1e-6, 'bwd_bias')


def test_bce_bits(pkg):
    from linr_pcgc_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    n = 50021
    z = torch.randn(n, generator=g) * 6
    z[:6] = torch.tensor([120.0, -120.0, 30.0, -30.0, 0.0, 17.5])       # saturating sigmoid, clamped logs
    t8 = (torch.rand(n, 8, generator=g) < 0.4).float()
    t8[:6, 3] = torch.tensor([0.0, 1.0, 0.0, 1.0, 1.0, 0.0])
    zo = z.clone().requires_grad_()
    ref_bits = torch.nn.functional.binary_cross_entropy(torch.sigmoid(zo), t8[:, 3], reduction='sum') / math.log(2.0)
    ref_bits.backward()
    p, bits = ops.bce_bits_fwd(z.to(dev), t8.to(dev)[:, 3])
    _close(p, torch.sigmoid(z), 1e-6, 1e-7, 'sigmoid')
    assert abs(float(bits) - float(ref_bits)) <= 1e-5 * float(ref_bits)
    gz = ops.bce_bits_bwd(p, t8.to(dev)[:, 3], 1.0 / math.log(2.0))
    _close(gz, zo.grad, 1e-4, 1e-6, 'bce bwd')


def test_adam_matches_torch(pkg):
    from linr_pcgc_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(54712, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    p, m, v = p0.to(dev), torch.zeros(54712, device=dev), torch.zeros(54712, device=dev)
    for step in range(1, 6):
        gr = torch.randn(54712, generator=g) * (0.1 ** step)
        ref.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.to(dev), m, v, step, 0.01)
        _close(p, ref, 2e-6, 2e-7, 'adam step %d' % step)


# ---- executor kernels through their own C-ABI entry ------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout', [(8, 8), (8, 4), (4, 4), (3, 8), (7, 8)])
def test_cmap_mfma_conv_bit_identical_to_gather_kernel(pkg, shell, cin, cout):
    """Compressed kernel map + v_mfma_f32_4x4x1 must reproduce the plain gather kernel bit for bit (K = 1 MFMA = fmaf)."""
    from linr_pcgc_amd import ops
    dev = _dev()
    sc = shell['scales'][0]
    n = len(sc['coord'])
    g = torch.Generator().manual_seed(cin * 17 + cout)
    nbr = ops.kmap_build(torch.from_numpy(sc['coord']).to(dev))
    lo, mask = ops.kmap_compress(nbr)
    # the compressed map must decode to the same table
    dec = torch.full_like(nbr, -1)
    for q in range(9):
        b0, b1, b2 = (mask >> (3 * q)) & 1, (mask >> (3 * q + 1)) & 1, (mask >> (3 * q + 2)) & 1
        dec[q] = torch.where(b0 == 1, lo[q], dec[q])
        dec[q + 9] = torch.where(b1 == 1, lo[q] + b0, dec[q + 9])
        dec[q + 18] = torch.where(b2 == 1, lo[q] + b0 + b1, dec[q + 18])
    assert torch.equal(dec, nbr)
    ld_in = 8
    xb = torch.zeros((n + 1, ld_in), device=dev)
    xb[1:, :cin] = torch.randn(n, cin, generator=g).to(dev)
    w = (torch.randn(27, cin, cout, generator=g) * 0.2).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    ref = ops.spconv_fwd(xb[1:], nbr, w, b.view(1, -1), relu=True, pad_row=True)
    got = ops.spconv_cmap(xb[1:], lo, mask, n, w, b, relu=True)
    assert torch.equal(ref, got)
    if cin in (4, 8):                       # backward-data: gathered width cout, produced width cin
        gb = torch.zeros((n + 1, 8), device=dev)
        gb[1:, :cout] = torch.randn(n, cout, generator=g).to(dev)
        ref_b = ops.spconv_bwd_data(gb[1:], nbr, w, pad_row=True)
        got_b = ops.spconv_cmap(gb[1:], lo, mask, n, w, None, bwd=True)
        assert torch.equal(ref_b, got_b)


@pytest.mark.parametrize('cin,cout', [(8, 8), (8, 4), (3, 8)])
def test_wgrad_cmap_entry_matches_oracle(pkg, shell, cin, cout):
    """linr_spconv_wgrad_cmap (the executor's backward-weight kernel) against autograd of the oracle convolution."""
    from linr_pcgc_amd import ops
    dev = _dev()
    sc = shell['scales'][0]
    n = len(sc['coord'])
    g = torch.Generator().manual_seed(31 * cin + cout)
    x = torch.randn(n, cin, generator=g)
    go = torch.randn(n, cout, generator=g)
    wo = (torch.randn(27, cin, cout, generator=g) * 0.2).requires_grad_()
    bo = torch.zeros(1, cout, requires_grad=True)
    onet.conv3(x, torch.from_numpy(sc['nbr']).long(), wo, bo).backward(go)
    nbr = ops.kmap_build(torch.from_numpy(sc['coord']).to(dev))
    xb = torch.zeros((n + 1, 8), device=dev)
    xb[1:, :cin] = x.to(dev)
    gw, gb = ops.spconv_wgrad_cmap(xb[1:], go.to(dev), nbr, n, cin, cout)                      # direct gathers, indices from nbr
    _close(gw, wo.grad, 0, 1e-4 * float(wo.grad.abs().max()) + 1e-6, 'wgrad cmap')
    _close(gb, bo.grad.reshape(-1), 0, 1e-4 * float(bo.grad.abs().max()) + 1e-6, 'bias grad cmap')
    slab1 = ops.spconv_wgrad_cmap(xb[1:], go.to(dev), nbr, n, cin, cout, reduce=False)
    # 16-byte friendly leading dimension (16-byte index loads) and the transposing kernel: bit-identical partials
    ld4 = (n + 63) // 64 * 64
    nbr4 = torch.full((27, ld4), -1, dtype=torch.int32, device=dev)
    nbr4[:, :n] = nbr
    slab_t = ops.spconv_wgrad_cmap(xb[1:], go.to(dev), nbr4, n, cin, cout, reduce=False)
    assert torch.equal(slab1, slab_t), 'scalar and 16-byte index loads must give the same partials'
    slab_tt = ops.spconv_wgrad_cmap(xb[1:], go.to(dev), nbr4, n, cin, cout, reduce=False, tile8t=ops.kmap_tile8t(nbr4, n))
    assert torch.equal(slab_t, slab_tt), 'coalesced-gather + LDS-transpose weight gradients must equal the table kernel bit for bit'
    slab2 = ops.spconv_wgrad_cmap(xb[1:], go.to(dev), nbr, n, cin, cout, reduce=False)
    assert torch.equal(slab1, slab2), 'partials must be bit-reproducible'


@pytest.mark.parametrize('n', [1, 2, 85, 256, 4099, 800001])
def test_coords_minmax_matches_torch(pkg, n):
    """linr_coords_minmax against the column reductions of custom_dataset.py:276-279 (negative coordinates included), bit-exact."""
    from linr_pcgc_amd import ops
    g = torch.Generator().manual_seed(n)
    c = torch.randint(-1000, 1 << 20, (n, 3), generator=g, dtype=torch.int32)
    c[:, 1] += 77
    got = ops.coords_minmax(c.to(_dev())).cpu()
    assert got[:3].tolist() == c.min(dim=0).values.tolist() and got[3:].tolist() == c.max(dim=0).values.tolist()


@pytest.mark.parametrize('n,span,shift', [(1, 4, 0), (5000, 40, 0), (5000, 40, 1), (200000, 1000, 1), (70000, (1 << 20) - 1, 0), (0, 4, 1)])
def test_coords_sort_unique_matches_torch_unique(pkg, n, span, shift):
    """linr_coords_sort_unique (input de-dup and the parent step of every octree level as one call) against torch.unique(dim=0)
    of (coords >> shift): same rows, same x-major order, bit-exact (custom_dataset.py:271-282, module_utils.py:92,103)."""
    from linr_pcgc_amd import ops
    g = torch.Generator().manual_seed(n + span + shift)
    c = torch.randint(0, span + 1, (n, 3), generator=g, dtype=torch.int64)
    if n > 10:
        c[: n // 3] = c[n // 3: 2 * (n // 3)]                  # plenty of duplicates
    got = ops.coords_sort_unique(c.to(torch.int32).cuda().contiguous(), shift, max(1, int(span).bit_length())).cpu()
    if n:
        assert torch.equal(ops.coords_sort_unique(c.to(torch.int32).cuda().contiguous(), shift).cpu(), got)      # default: 20-bit keys
    want = torch.unique(c >> shift, dim=0).to(torch.int32) if n else torch.zeros((0, 3), dtype=torch.int32)
    assert got.shape == want.shape and torch.equal(got, want)


def test_octree_levels_through_the_library_match_the_oracle(pkg):
    """prepare_frame on the GPU (sort + unique + occupancy in the library) against the numpy oracle, every scale of a sphere shell:
    coordinates, child occupancy and the point count bit-exact."""
    from linr_pcgc_amd import synthetic
    from linr_pcgc_amd.module_utils import prepare_frame
    pts = synthetic.sphere_shell(7, 50.0)
    rng = np.random.default_rng(3)
    pts = np.concatenate([pts, pts[rng.integers(0, len(pts), 500)]])[rng.permutation(len(pts) + 500)]      # shuffled, with duplicates
    pts = pts + np.array([5, -70, 1000], dtype=pts.dtype)                      # a frame minimum that is not the origin (coord_data_min)
    fr = prepare_frame(torch.as_tensor(pts).cuda(), None, 64, device='cuda', with_offsets=False)
    ref = ooct.prepare_frame(pts, None, 64)
    assert fr['coord_data_min'] == [int(v) for v in ref['coord_data_min']]
    assert np.array_equal(fr['ori'].cpu().numpy(), ref['ori'])
    fr64 = prepare_frame(torch.as_tensor(pts.astype(np.int64)).cuda(), None, 64, device='cuda', with_offsets=False)      # int64 input: same result
    assert torch.equal(fr64['ori'], fr['ori']) and fr64['coord_data_min'] == fr['coord_data_min']
    assert fr['point_num'] == ref['point_num'] and fr['scale_num'] == ref['scale_num']
    for a, b in zip(fr['all_input_info'], ref['scales']):
        assert np.array_equal(a['coord'].cpu().numpy(), b['coord'])
        assert np.array_equal(a['occ'].cpu().numpy(), b['occ'])


def _frames_equal(a, b):
    assert a['point_num'] == b['point_num'] and a['scale_num'] == b['scale_num'] and a['coord_data_min'] == b['coord_data_min']
    assert torch.equal(a['ori'], b['ori'])
    for x, y in zip(a['all_input_info'], b['all_input_info']):
        assert x['scale_idx'] == y['scale_idx']
        assert torch.equal(x['coord'], y['coord']) and torch.equal(x['occ'], y['occ']) and torch.equal(x['ground_truth'], y['ground_truth'])


@pytest.mark.parametrize('case', ['sphere7', 'rough8', 'random6', 'tiny', 'scale_num_2', 'deep'])
def test_all_levels_in_one_call_equal_the_per_level_entry(pkg, case, monkeypatch):
    """linr_octree_levels (csrc/octree.hip: every level of a frame in one call - parents as the set bits of a key bitmap, no sort, counts
    chained on the device, one host read) against linr_octree_level applied level by level (LINR_OCTREE_PER_LEVEL=1) and against the
    numpy oracle: coordinates, child occupancy, counts, the stop criterion (min_point_num, a given scale_num, levels that run out of
    coordinate bits) - bit-exact."""
    from linr_pcgc_amd import synthetic
    from linr_pcgc_amd.module_utils import prepare_frame
    rng = np.random.default_rng(11)
    scale_num, minp = None, 64
    if case == 'sphere7':
        pts = synthetic.sphere_shell(7, 50.0) + np.array([3, -9, 40], dtype=np.int32)
    elif case == 'rough8':
        pts = synthetic.rough_figure(8, 3)
    elif case == 'random6':
        pts = rng.integers(0, 64, size=(30000, 3)).astype(np.int32)           # dense random cloud with duplicates, unsorted
    elif case == 'tiny':
        pts = np.array([[5, 5, 5], [5, 5, 6], [9, 1, 0]], dtype=np.int32)
        minp = 1
    elif case == 'scale_num_2':
        pts = synthetic.sphere_shell(7, 50.0)
        scale_num = 2
    else:                                                                      # 'deep': more levels asked for than coordinate bits
        pts = synthetic.sphere_shell(5, 9.0)
        scale_num, minp = 9, 0
    dev_pts = torch.as_tensor(pts).cuda()
    fast = prepare_frame(dev_pts, scale_num, minp, device='cuda', with_offsets=False)
    monkeypatch.setenv('LINR_OCTREE_PER_LEVEL', '1')
    slow = prepare_frame(dev_pts, scale_num, minp, device='cuda', with_offsets=False)
    monkeypatch.delenv('LINR_OCTREE_PER_LEVEL')
    _frames_equal(fast, slow)
    ref = ooct.prepare_frame(pts, scale_num, minp)
    assert fast['scale_num'] == ref['scale_num'] and fast['point_num'] == ref['point_num']
    for a, b in zip(fast['all_input_info'], ref['scales']):
        assert np.array_equal(a['coord'].cpu().numpy(), b['coord']) and np.array_equal(a['occ'].cpu().numpy(), b['occ'])


def test_all_levels_in_one_call_at_config4_size(pkg, monkeypatch):
    """... and at BASELINE config[4]'s frame size (11-bit, ~2.9 M points: the 128 MB bitmap of the finest parent level)."""
    from linr_pcgc_amd import synthetic
    from linr_pcgc_amd.module_utils import prepare_frame
    pts = synthetic.sequence_frame_device('owlii11', 2, 'cuda')
    fast = prepare_frame(pts, None, 64, device='cuda', with_offsets=False)
    monkeypatch.setenv('LINR_OCTREE_PER_LEVEL', '1')
    slow = prepare_frame(pts, None, 64, device='cuda', with_offsets=False)
    _frames_equal(fast, slow)
    assert fast['scale_num'] == 8 and fast['point_num'] > 2800000


def test_sphere_generator_on_the_gpu_equals_the_numpy_one(pkg):
    """synthetic.sphere_shell_device (candidate windows around the two roots per (x, y) column) gives sphere_shell's voxel list, row for
    row, also at the equator, for thick shells, tiny radii and off-centre spheres."""
    from linr_pcgc_amd import synthetic
    for b, r, c, th in [(6, 20, None, 0.5), (7, 40, (60, 64, 70), 1.0), (8, 100, None, 0.5), (8, 100, (130, 128, 120), 1.0), (7, 5, None, 0.5),
                        (8, 3, None, 2.5), (8, 90, None, 3.0), (7, 2, None, 0.5)]:
        cc = [(1 << b) // 2] * 3 if c is None else c
        assert np.array_equal(synthetic.sphere_shell(b, r, cc, th), synthetic.sphere_shell_device(b, r, cc, th, 'cuda').cpu().numpy()), (b, r, c, th)
    assert synthetic.sequence_frame_device('loot10', 0, 'cuda').shape[0] == 784314
