"""Oracle (test infrastructure): torch-CPU emulation of the bf16 / uint8-weight inference path (csrc/net_bf16.hip).

Same wiring as oracle/network.py (models/model_core.py:38-81, models/upsample.py:137-217, models/resnet.py:55-60), with
the roundings of the bf16 executor put where it puts them: 3x3x3 kernels rounded to bf16, every activation that the
executor stores (x_low, A, H, M, I, O / x_glob) rounded to bf16, products accumulated in fp32, biases / pointwise convs /
MLPs in fp32, the prune-conv output fed to the head MLP WITHOUT rounding.  It differs from the HIP path only in the order of
the fp32 additions (and the bf16 roundings that order can flip), so it pins the HIP result far tighter than the fp32
oracle can (tests/test_gpu_bf16.py states both tolerances).  PARITY UNPINNED against the reference, which has no
reduced-precision path at all (models/quantize_functions.py quantises coordinates; its 8-bit weight code is
model_compression/model_size_est.py:72-91, which oracle/model_codec.py restates and the golden side_info.json pins).
"""
import torch
import torch.nn.functional as F

from . import network as onet


def rb(x):
    """round to bf16 (RNE), keep float32 storage"""
    return x.to(torch.bfloat16).to(torch.float32)


def conv3(x, nbr, kernel, bias):
    return onet.conv3(x, nbr, rb(kernel), bias)


def inception(x, nbr, sd, p):
    h0 = F.relu(conv3(x, nbr, sd[p + '.conv0_0.kernel'], sd[p + '.conv0_0.bias']))
    h1 = F.relu(onet.conv1(x, sd[p + '.conv1_0.kernel'], sd[p + '.conv1_0.bias']))
    h = rb(torch.cat([h0, h1], dim=1))
    out0 = conv3(h[:, :4], nbr, sd[p + '.conv0_1.kernel'], sd[p + '.conv0_1.bias'])
    m = rb(F.relu(conv3(h[:, 4:], nbr, sd[p + '.conv1_1.kernel'], sd[p + '.conv1_1.bias'])))
    out1 = onet.conv1(m, sd[p + '.conv1_2.kernel'], sd[p + '.conv1_2.bias'])
    return rb(torch.cat([out0, out1], dim=1) + x)


def make_block(x, nbr, sd, p, res=None):
    a = rb(F.relu(conv3(x, nbr, sd[p + '.0.kernel'], sd[p + '.0.bias'])))
    out, nl = a, 0
    while (p + '.2.layers.%d.conv0_0.kernel' % nl) in sd:
        out = inception(out, nbr, sd, p + '.2.layers.%d' % nl)
        nl += 1
    if nl > 1:
        out = rb(out + a)
    o = conv3(out, nbr, sd[p + '.3.kernel'], sd[p + '.3.bias'])
    return rb(o if res is None else o + res)


def forward_scale(sd, scale):
    """logits / probs of one scale dict {'offset_tensor','occ','nbr','scale_idx'}; sd = the DE-QUANTISED fp32 state dict."""
    u = 'upsampler.'
    nbr, occ = scale['nbr'], scale['occ']
    x_low = rb(onet.scale_context(sd, scale['offset_tensor'], scale['scale_idx']))
    x_glob = make_block(x_low, nbr, sd, u + 'block_in')
    logits, probs = [], []
    prior = x_glob
    for k in range(8):
        c = conv3(prior, nbr, sd[u + 'prune_blocks.%d.0.conv.kernel' % k], sd[u + 'prune_blocks.%d.0.conv.bias' % k])
        z = onet.mlp(c, sd, u + 'inner_mlps.%d.0' % k)
        logits.append(z)
        probs.append(torch.sigmoid(z))
        if k == 7:
            break
        prior = make_block(occ[:, :k + 1], nbr, sd, u + 'outter_blocks.%d' % k, res=x_glob)
    return {'logits': logits, 'probs': probs, 'bits': onet.bits_of(probs, occ)}


# ---- bf16 TRAINING executor (csrc/train_bf16.hip): autograd-capable emulation -------------------------------------------------------
# Same wiring (models/model_core.py:38-81, models/upsample.py:88-97,137-217, models/resnet.py:12-60; main.py:315-316 differentiates
# it), with that executor's ONE rounding rule: every matrix it writes to memory is rounded to bf16 and every consumer sees the stored
# value - forward (x_low, A, H, M, I, O / x_glob, the prune convolutions' outputs C) AND backward (the gradient that arrives at each
# of those tensors is rounded once, after all its contributions were summed in fp32).  3x3x3 kernels are rounded to bf16 where they
# are multiplied (forward and backward-data) while their gradient goes to the fp32 master unrounded; everything else is fp32.
# Differs from the HIP path only in the order of fp32 additions (and the bf16 roundings that order can flip).  PARITY UNPINNED
# against the reference, which trains in fp32 only.

class _StoreBf16(torch.autograd.Function):
    """y = bf16(x) going forward, g_x = bf16(g_y) going backward: a tensor the executor keeps in memory."""

    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


def st(x):
    return _StoreBf16.apply(x)


def wq(w):
    """bf16 value of a kernel, gradient straight to the fp32 master"""
    return w + (rb(w) - w).detach()


def conv3t(x, nbr, kernel, bias):
    return onet.conv3(x, nbr, wq(kernel), bias)


def inception_t(x, nbr, sd, p):
    h0 = F.relu(conv3t(x, nbr, sd[p + '.conv0_0.kernel'], sd[p + '.conv0_0.bias']))
    h1 = F.relu(onet.conv1(x, sd[p + '.conv1_0.kernel'], sd[p + '.conv1_0.bias']))
    h = st(torch.cat([h0, h1], dim=1))
    out0 = conv3t(h[:, :4], nbr, sd[p + '.conv0_1.kernel'], sd[p + '.conv0_1.bias'])
    m = st(F.relu(conv3t(h[:, 4:], nbr, sd[p + '.conv1_1.kernel'], sd[p + '.conv1_1.bias'])))
    out1 = onet.conv1(m, sd[p + '.conv1_2.kernel'], sd[p + '.conv1_2.bias'])
    return st(torch.cat([out0, out1], dim=1) + x)


def make_block_t(x, nbr, sd, p, res=None):
    a = st(F.relu(conv3t(x, nbr, sd[p + '.0.kernel'], sd[p + '.0.bias'])))
    if (p + '.2.layers.1.conv0_0.kernel') in sd:
        raise ValueError('the bf16 training executor supports block_layers=1 only')
    i = inception_t(a, nbr, sd, p + '.2.layers.0')
    o = conv3t(i, nbr, sd[p + '.3.kernel'], sd[p + '.3.bias'])
    return st(o if res is None else o + res)


def train_forward_scale(sd, scale):
    """Differentiable forward of one scale dict {'offset_tensor','occ','nbr','scale_idx'}; sd = the fp32 MASTER state dict
    (tensors with requires_grad for gradients)."""
    u = 'upsampler.'
    nbr, occ = scale['nbr'], scale['occ']
    x_low = st(onet.scale_context(sd, scale['offset_tensor'], scale['scale_idx']))
    x_glob = make_block_t(x_low, nbr, sd, u + 'block_in')
    logits, probs = [], []
    prior = x_glob
    for k in range(8):
        c = st(conv3t(prior, nbr, sd[u + 'prune_blocks.%d.0.conv.kernel' % k], sd[u + 'prune_blocks.%d.0.conv.bias' % k]))
        z = onet.mlp(c, sd, u + 'inner_mlps.%d.0' % k)
        logits.append(z)
        probs.append(torch.sigmoid(z))
        if k == 7:
            break
        prior = make_block_t(occ[:, :k + 1], nbr, sd, u + 'outter_blocks.%d' % k, res=x_glob)
    return {'logits': logits, 'probs': probs, 'bits': onet.bits_of(probs, occ)}


def train_frame_bits(sd, scales):
    """main.overfit_one_frame (main.py:457-475) on the emulated executor: sum of per-scale bits (differentiable)."""
    total = 0
    for s in scales:
        total = total + train_forward_scale(sd, s)['bits']
    return total
