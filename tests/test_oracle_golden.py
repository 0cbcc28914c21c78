"""CPU: the oracle against the reference-generated golden vectors and the shipped loot/ known answers."""
import os

import numpy as np
import pytest
import torch

from oracle import ac, model_codec, octree


@pytest.mark.parametrize('name', ['octree_random64.npz', 'octree_shell128.npz'])
def test_octree_prep_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    frame = octree.prepare_frame(g['points'], None, int(g['min_point_num']))
    assert frame['scale_num'] == int(g['scale_num'])
    assert (frame['coord_data_min'] == g['coord_data_min']).all()
    assert (frame['ori'] == g['ori']).all()
    for s, sc in enumerate(frame['scales']):
        assert (sc['coord'] == g['s%d_coord' % s]).all()
        assert (sc['occ'] == g['s%d_occ' % s]).all()
        assert (sc['offset_tensor'] == g['s%d_offset' % s]).all()
        assert (octree.upper_layer(sc['coord'], sc['occ']) == g['s%d_upper' % s]).all()
        assert (octree.upper_layer(sc['coord'], sc['occ']) == sc['ground_truth']).all()


def test_neighbour_table_properties(golden_dir):
    g = np.load(os.path.join(golden_dir, 'octree_shell128.npz'))
    c = g['s0_coord']
    nbr = octree.neighbour_table(c)
    assert (nbr[:, 13] == np.arange(len(c))).all()
    for k in range(27):
        d = np.array([k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1])
        hit = nbr[:, k] >= 0
        assert (c[nbr[hit, k]] == c[hit] + d).all()
        # mirror identity used by the gather-form backward: nbr[nbr[j,k], 26-k] == j
        assert (nbr[nbr[hit, k], 26 - k] == np.nonzero(hit)[0]).all()
    # 7-neighbour occupancy feature == presence of the axis neighbours in the table
    off = g['s0_offset']
    for col, k in zip(range(7), [13, 12, 14, 10, 16, 4, 22]):
        assert ((nbr[:, k] >= 0) == (off[:, col] == 1)).all()


def test_quantiser_known_answer(golden_dir):
    g = np.load(os.path.join(golden_dir, 'loot_model_kat.npz'))
    q, recon, mn, mx = model_codec.quant_uniform2(g['flat'], int(g['bitdepth']))
    mu, b = model_codec.laplace_params(q)
    assert float(mu) == float(g['mu']) and float(b) == float(g['b'])
    assert float(mn) == float(g['min_param']) and float(mx) == float(g['max_param'])
    assert q.min() == 0 and q.max() == 255


def test_model_stream_known_answer(golden_dir):
    """Known-answer test of the model stream (torchac restatement + Laplace CDF quirk of model_size_est.py:466-482) on the
    shipped checkpoint - stated as what is computed, no more.
    The artefact: loot/gop_32_62/70/result.json's model_bpp, bpp_t and xyzlow_bpp are integral bit counts only for
    P = 24,372,190 points, giving 282,642 model bits = 8 L + header bits.
    Computed here:
      * symbols, mu = 128, b = 6, min / max reproduce side_info.json exactly (test_quantiser_known_answer);
      * with the pdf exp(-|x - mu| / b) / (2 b) evaluated by the CPU's expf, the 257-entry integer CDF is the same under
        sequential-fp32, fp64 and log-step-scan accumulation, and torchac's published coder emits L = 35,319 bytes;
      * BUT the reference evaluates that pdf on CUDA (model_size_est.py:470-476, device=mu.device), whose expf / division may
        differ from the CPU's in the last bit, and ten CDF entries sit within 0.02 of a rounding boundary of cdf * 65280
        (entries 143 and 194 exactly ON .5 in fp32, where round-half-to-even decides).  Moving single entries across their
        nearest boundary: 194 alone, 177 alone, or {143, 194} give L = 35,320; {143, 177, 194} gives 35,321; each of the
        other seven alone leaves 35,319.
    282,642 = 8 * 35,319 + 90 = 8 * 35,320 + 82.  Every header formula in today's tree is 82 bits (model_size_est.py:166,247,
    448,484,489).  So TWO explanations are consistent with the artefact and this test cannot tell them apart offline:
      (a) CPU-exact pdf, L = 35,319, written by an older revision with a 90-bit header (the artefact's keys bpp_t /
          fake_bpp_all are no longer written by test_utils.py:157), or
      (b) today's 82-bit header and L = 35,320, from a CUDA pdf that differs from the CPU's in the last bit of a few entries.
    Corollary for the format: a model.bin written with a CUDA-built CDF need not decode with a CPU-built one (and vice
    versa) - the Laplace stream is only portable between devices whose expf agree to the last bit on these 256 values.
    What the oracle is pinned to: the coder and the CDF quirk (a coder / CDF defect moves L by many bytes: without the
    quirk L = 34,941), and L = 35,319 for the CPU pdf."""
    g = np.load(os.path.join(golden_dir, 'loot_model_kat.npz'))
    P = 24372190
    for key in ('model_bpp', 'bpp_t', 'xyzlow_bpp'):
        bits = float(g[key]) * P
        assert abs(bits - round(bits)) < 1e-6
    model_bits = round(float(g['model_bpp']) * P)
    assert model_bits == 282642 == 8 * 35319 + 90 == 8 * 35320 + 82
    out = model_codec.encode_model(g['flat'], 8)
    assert len(out['bytes']) == 35319                       # CPU pdf
    # the integer CDF does not depend on how the float CDF was accumulated ...
    q, _, _, _ = model_codec.quant_uniform2(g['flat'], 8)
    mu, b = model_codec.laplace_params(q)
    x = torch.arange(256.0)
    pdf = torch.exp(-torch.abs(x - mu) / b) / (2 * b)
    pdf = pdf / pdf.sum()

    def to_int(cdf):
        return ac.cdf_float_to_int(torch.cat([cdf.to(torch.float32), torch.zeros(1)]).numpy()[None, :])

    seq = to_int(torch.cumsum(pdf, dim=-1))
    f64 = to_int(torch.cumsum(pdf.double(), dim=-1))
    scan = pdf.clone()
    d = 1
    while d < 256:                                   # Hillis-Steele scan in fp32: the association a GPU cumsum uses
        nxt = scan.clone()
        nxt[d:] = scan[d:] + scan[:-d]
        scan, d = nxt, 2 * d
    assert np.array_equal(seq, f64) and np.array_equal(seq, to_int(scan))
    # ... but it does depend on the last bit of the pdf: the entries near a rounding boundary and what their flips do to L
    cdf = model_codec.laplace_cdf(mu, b, 8).numpy()
    scaled = cdf.astype(np.float32) * np.float32(65280)
    frac = scaled - np.floor(scaled)
    near = [i for i in range(256) if abs(float(frac[i]) - 0.5) < 0.02]
    assert near == [61, 78, 95, 112, 119, 136, 143, 160, 177, 194]
    assert float(frac[143]) == 0.5 and float(frac[194]) == 0.5
    base = seq[0].astype(np.int64)
    sym = q.numpy().astype(np.int16)

    def length(flips):
        c = base.copy()
        for i in flips:
            c[i] += 1 if frac[i] < 0.5 else -1       # across the entry's nearest rounding boundary
        return len(ac.encode_int_cdf(np.broadcast_to((c & 0xFFFF).astype(np.uint16), (len(sym), 257)), sym))

    single = {i: length([i]) for i in near}
    assert single == {61: 35319, 78: 35319, 95: 35319, 112: 35319, 119: 35319, 136: 35319, 143: 35319, 160: 35319,
                      177: 35320, 194: 35320}
    assert length([143, 194]) == 35320 and length([143, 177, 194]) == 35321
    assert 8 * length([143, 194]) + 82 == model_bits       # explanation (b)
    assert 8 * length([]) + 90 == model_bits               # explanation (a)
    rec, sym_d = model_codec.decode_model(out['bytes'], len(g['flat']), out['mu'], out['b'], out['min_param'],
                                          out['max_param'])
    assert (sym_d.astype(np.uint8) == out['symbols']).all()
    assert torch.equal(rec, out['recon'])


def test_binary_coder_roundtrip_and_rate():
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 17, 50000):
        p = rng.random(n).astype(np.float32)
        if n > 4:
            p[:4] = [0.0, 1.0, 1e-9, 1 - 1e-7]          # saturated probabilities (sigmoid in fp32)
        s = (rng.random(n) < p).astype(np.int16)
        if n > 4:
            s[:4] = [1, 0, 1, 0]                         # worst case: the "impossible" symbol still decodes
        data = ac.encode_binary(p, s)
        assert (ac.decode_binary(p, data) == s).all()
        if n == 50000:
            c1 = ac.cdf_float_to_int(ac.binary_cdf(p))[:, 1].astype(np.float64)
            ideal = -np.log2(np.where(s == 1, 65536 - c1, c1) / 65536).sum()
            assert ideal <= len(data) * 8 <= ideal + 16


def test_cdf_conversion_binary_bounds():
    c = ac.cdf_float_to_int(ac.binary_cdf(np.array([0.0, 1.0, 0.5], dtype=np.float32)))
    assert c[:, 0].tolist() == [0, 0, 0]
    assert c[:, 1].tolist() == [65535, 1, 32768]
    assert c[:, 2].tolist() == [0, 0, 0]               # 65534 + 2 wraps; never read (c_high is hard-wired 2^16)


def _reference_state_dict(golden_dir, perm=(0, 1, 2), mirrored=False):
    """The reference-trained checkpoint (loot/gop_32_62/model.pth, shipped with the reference) with its 27 kernel taps
    re-read under a candidate offset convention: tap k = (d[perm0]+1) + 3 (d[perm1]+1) + 9 (d[perm2]+1), d -> -d if
    mirrored.  perm (0,1,2) unmirrored is the convention the oracle (and the HIP kernels) assume for MinkowskiEngine."""
    import ast
    g = np.load(os.path.join(golden_dir, 'loot_model_kat.npz'), allow_pickle=True)
    flat = torch.from_numpy(g['flat'].astype(np.float32))
    sd, off = {}, 0
    for name, shape in zip(g['names'], g['shapes']):
        shape = ast.literal_eval(str(shape))
        cnt = int(np.prod(shape))
        w = flat[off:off + cnt].view(shape).clone()
        off += cnt
        if w.dim() == 3 and w.shape[0] == 27:
            idx = []
            for k in range(27):
                d = [k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1]
                if mirrored:
                    d = [-v for v in d]
                idx.append((d[perm[0]] + 1) + 3 * (d[perm[1]] + 1) + 9 * (d[perm[2]] + 1))
            w = w[idx]
        sd[str(name)] = w
    assert off == flat.numel()
    return sd


def test_reference_checkpoint_pins_network_semantics(golden_dir):
    """Behavioural pin of the network restatement (MinkowskiEngine itself is not available): a model the REFERENCE trained
    on real loot must also predict the occupancy of an unseen smooth surface - but only if the restatement applies its
    weights the way MinkowskiEngine did (tap order, correlation direction, block wiring, channel order of the occupancy
    concat, child index convention).  Measured: 0.96 bits/point under the assumed convention, 6.6-10.8 under each of the
    11 other axis orders / mirrorings, 3.4 for an untrained network."""
    import itertools
    from oracle import network as onet
    g = np.load(os.path.join(golden_dir, 'octree_shell128.npz'))
    scales = []
    for s in range(int(g['scale_num'])):
        c = g['s%d_coord' % s]
        scales.append({'coord': c, 'occ': g['s%d_occ' % s], 'offset_tensor': g['s%d_offset' % s], 'scale_idx': s,
                       'nbr': octree.neighbour_table(c)})
    tsc = onet.to_torch_scales(scales)
    points = len(g['ori'])
    bpp = {}
    with torch.no_grad():
        for perm in itertools.permutations(range(3)):
            for mirrored in (False, True):
                bpp[(perm, mirrored)] = float(onet.frame_bits(_reference_state_dict(golden_dir, perm, mirrored), tsc)) / points
    assumed = bpp[((0, 1, 2), False)]
    assert assumed < 1.2, bpp
    others = [v for k, v in bpp.items() if k != ((0, 1, 2), False)]
    assert min(others) > 4.0 * assumed, bpp


def test_reference_checkpoint_pins_wiring_and_channel_conventions(golden_dir, monkeypatch):
    """The same behavioural pin for the conventions the tap-order sweep does not touch.  The network the REFERENCE trained is run
    through the restatement as written and through eleven variants that each read one convention the other plausible way; every
    variant predicts the unseen surface worse (bits/point, measured: assumed 0.963):
      child index 4dz+2dy+dx instead of 4dx+2dy+dz (module_utils.py:93) 6.73 | child order reversed 10.89 | occupancy concat newest
      first (upsample.py:206-209) 3.54 | Inception cat([out1, out0]) (resnet.py:55-60) 3.68 | prior_k on the previous prior instead of
      the ORIGINAL x_glob (upsample.py:213) 5.74 | scale_idx counted from the coarsest 2.05 | scale_idx off by one 1.54 | scale
      context [offsets | emb] instead of [emb | offsets] (model_core.py:48-53) 1.41 | 7-neighbour offsets x<->z 1.24, +<->- 1.13
      (glob_params.py:3) | make_block without its ReLU (upsample.py:88-97) 1.16."""
    import torch.nn.functional as F
    from oracle import network as onet
    g = np.load(os.path.join(golden_dir, 'octree_shell128.npz'))
    scales = []
    for s in range(int(g['scale_num'])):
        c = g['s%d_coord' % s]
        scales.append({'coord': c, 'occ': g['s%d_occ' % s], 'offset_tensor': g['s%d_offset' % s], 'scale_idx': s,
                       'nbr': octree.neighbour_table(c)})
    points, S = len(g['ori']), len(scales)
    sd = _reference_state_dict(golden_dir, (0, 1, 2), False)

    def bpp(sc):
        with torch.no_grad():
            return float(onet.frame_bits(sd, onet.to_torch_scales(sc))) / points
    assumed = bpp(scales)
    assert assumed < 1.2
    # conventions of the inputs
    dz_major = [((j >> 2) & 1) | (((j >> 1) & 1) << 1) | ((j & 1) << 2) for j in range(8)]
    data_variants = {
        'child index dz-major': ([dict(s, occ=s['occ'][:, dz_major]) for s in scales], 3.0),
        'child order reversed': ([dict(s, occ=s['occ'][:, ::-1].copy()) for s in scales], 3.0),
        'offsets x<->z': ([dict(s, offset_tensor=s['offset_tensor'][:, [0, 5, 6, 3, 4, 1, 2]]) for s in scales], 1.1),
        'offsets +<->-': ([dict(s, offset_tensor=s['offset_tensor'][:, [0, 2, 1, 4, 3, 6, 5]]) for s in scales], 1.1),
        'scale_idx from the coarsest': ([dict(s, scale_idx=S - 1 - s['scale_idx']) for s in scales], 1.5),
        'scale_idx off by one': ([dict(s, scale_idx=min(s['scale_idx'] + 1, 6)) for s in scales], 1.3)}
    for name, (sc, factor) in data_variants.items():
        assert bpp(sc) > factor * assumed, name

    # conventions of the wiring: one restated function replaced at a time
    def inception_swapped(x, nbr, sdd, p):
        out0 = onet.conv3(F.relu(onet.conv3(x, nbr, sdd[p + '.conv0_0.kernel'], sdd[p + '.conv0_0.bias'])), nbr,
                          sdd[p + '.conv0_1.kernel'], sdd[p + '.conv0_1.bias'])
        h = F.relu(onet.conv1(x, sdd[p + '.conv1_0.kernel'], sdd[p + '.conv1_0.bias']))
        h = F.relu(onet.conv3(h, nbr, sdd[p + '.conv1_1.kernel'], sdd[p + '.conv1_1.bias']))
        return torch.cat([onet.conv1(h, sdd[p + '.conv1_2.kernel'], sdd[p + '.conv1_2.bias']), out0], dim=1) + x

    def cnp(cumulative, newest_first):
        def forward(sdd, x_low, occ, nbr, stages=8):
            u = 'upsampler.'
            x_glob = onet.make_block(x_low, nbr, sdd, u + 'block_in')
            logits, probs, prior = [], [], x_glob
            for k in range(stages):
                c = onet.conv3(prior, nbr, sdd[u + 'prune_blocks.%d.0.conv.kernel' % k], sdd[u + 'prune_blocks.%d.0.conv.bias' % k])
                z = onet.mlp(c, sdd, u + 'inner_mlps.%d.0' % k)
                logits.append(z)
                probs.append(torch.sigmoid(z))
                if k == stages - 1:
                    break
                seen = torch.flip(occ[:, :k + 1], dims=[1]) if newest_first else occ[:, :k + 1]
                prior = (prior if cumulative else x_glob) + onet.make_block(seen, nbr, sdd, u + 'outter_blocks.%d' % k)
            return logits, probs
        return forward

    def make_block_without_relu(x, nbr, sdd, p):
        out, nl = onet.conv3(x, nbr, sdd[p + '.0.kernel'], sdd[p + '.0.bias']), 0
        while (p + '.2.layers.%d.conv0_0.kernel' % nl) in sdd:
            out = onet.inception(out, nbr, sdd, p + '.2.layers.%d' % nl)
            nl += 1
        return onet.conv3(out, nbr, sdd[p + '.3.kernel'], sdd[p + '.3.bias'])

    def context_offsets_first(sdd, offset_tensor, scale_idx):
        emb = sdd['scale_emb.weight'][scale_idx].unsqueeze(0).expand(offset_tensor.shape[0], -1)
        return onet.mlp(torch.cat([offset_tensor, emb], dim=-1), sdd, 'scale_mlp.%d' % scale_idx)
    wiring_variants = {'inception': (inception_swapped, 2.5), 'cnp_forward': (cnp(True, False), 3.0), 'make_block': (make_block_without_relu, 1.1),
                       'scale_context': (context_offsets_first, 1.25)}
    for attr, (fn, factor) in wiring_variants.items():
        with monkeypatch.context() as mp:
            mp.setattr(onet, attr, fn)
            assert bpp(scales) > factor * assumed, attr
    with monkeypatch.context() as mp:
        mp.setattr(onet, 'cnp_forward', cnp(False, True))
        assert bpp(scales) > 2.5 * assumed, 'occupancy concat newest first'
    assert abs(bpp(scales) - assumed) < 1e-12          # the restatement is back as written
