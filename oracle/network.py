"""Oracle (test infrastructure): torch-CPU restatement of the LINR-PCGC coding network.

Dense-row form: every sparse tensor of one scale is a [N, C] matrix over the SAME N parent voxels (instage=1 makes
every octant mask all-True, upsample.py:33-34,99-109), every 3x3x3 convolution is
    out[j] = bias + sum_k  in[nbr[j, k]] @ kernel[k]          (absent neighbour -> zero row)
with nbr from oracle.octree.neighbour_table.  Functions take a plain ``state_dict`` (reference names,
loot/gop_32_62/model.pth loads unchanged) so autograd on those tensors yields the reference gradients.

Reference lines restated:
  scale context + bits        models/model_core.py:38-81
  CNP wiring                  models/upsample.py:137-217 (note :213 - always the ORIGINAL x_glob)
  make_block                  models/upsample.py:88-97
  InceptionResNet             models/resnet.py:7-60 ; ResNetBlock :146-162 (extra skip when block_layers > 1)
  PointwiseMLP                models/module_utils.py:42-81
  merge_two_frames (=concat)  models/function_utils.py:58-69
MinkowskiEngine semantics (third-party, not in tree): SURVEY.md Appendix B.  PARITY UNPINNED at bit level for this file;
the tap order / direction / wiring are confirmed by the reference-trained checkpoint (oracle/__init__.py).
"""
import math
import torch
import torch.nn.functional as F


def conv3(x, nbr, kernel, bias):
    """MinkowskiConvolution(kernel_size=3, stride=1) on a fixed coordinate set.  kernel [27,Cin,Cout], bias [1,Cout]."""
    n, cin = x.shape
    xp = torch.cat([x, x.new_zeros(1, cin)], dim=0)
    idx = torch.where(nbr < 0, torch.full_like(nbr, n), nbr)
    cols = xp[idx.reshape(-1)].reshape(n, 27 * cin)          # im2col over the kernel map
    return cols @ kernel.reshape(27 * cin, -1) + bias


def conv1(x, kernel, bias):
    """MinkowskiConvolution(kernel_size=1): kernel [Cin,Cout]."""
    return x @ kernel + bias


def mlp(x, sd, prefix):
    """PointwiseMLP([a, b, c]) = Linear, ReLU, Linear (module_utils.py:59-81)."""
    h = F.relu(F.linear(x, sd[prefix + '.0.weight'], sd[prefix + '.0.bias']))
    return F.linear(h, sd[prefix + '.2.weight'], sd[prefix + '.2.bias'])


def inception(x, nbr, sd, p):
    """resnet.py:55-60."""
    out0 = conv3(F.relu(conv3(x, nbr, sd[p + '.conv0_0.kernel'], sd[p + '.conv0_0.bias'])), nbr,
                 sd[p + '.conv0_1.kernel'], sd[p + '.conv0_1.bias'])
    h = F.relu(conv1(x, sd[p + '.conv1_0.kernel'], sd[p + '.conv1_0.bias']))
    h = F.relu(conv3(h, nbr, sd[p + '.conv1_1.kernel'], sd[p + '.conv1_1.bias']))
    out1 = conv1(h, sd[p + '.conv1_2.kernel'], sd[p + '.conv1_2.bias'])
    return torch.cat([out0, out1], dim=1) + x


def make_block(x, nbr, sd, p):
    """upsample.py:88-97: conv3 -> ReLU -> ResNetBlock -> conv3.  ResNetBlock (resnet.py:146-162) chains its Inception
    layers and adds its input once more when there is more than one; the layer count is read off the state dict."""
    a = F.relu(conv3(x, nbr, sd[p + '.0.kernel'], sd[p + '.0.bias']))
    out, nl = a, 0
    while (p + '.2.layers.%d.conv0_0.kernel' % nl) in sd:
        out = inception(out, nbr, sd, p + '.2.layers.%d' % nl)
        nl += 1
    if nl > 1:
        out = out + a
    return conv3(out, nbr, sd[p + '.3.kernel'], sd[p + '.3.bias'])


def scale_context(sd, offset_tensor, scale_idx):
    """model_core.py:48-53."""
    emb = sd['scale_emb.weight'][scale_idx]
    mix = torch.cat([emb.unsqueeze(0).expand(offset_tensor.shape[0], -1), offset_tensor], dim=-1)
    return mlp(mix, sd, 'scale_mlp.%d' % scale_idx)


def cnp_forward(sd, x_low, occ, nbr, stages=8):
    """CNP.forward (upsample.py:163-217) at outstage=8, instage=1.  occ [N,8] float {0,1}.

    Returns (logits list of [N,1], probs list of [N,1]).  Stage k>0 sees x_glob + outter_blocks[k-1](occ[:, :k]).
    """
    u = 'upsampler.'
    x_glob = make_block(x_low, nbr, sd, u + 'block_in')
    logits, probs = [], []
    prior = x_glob
    for k in range(stages):
        c = conv3(prior, nbr, sd[u + 'prune_blocks.%d.0.conv.kernel' % k], sd[u + 'prune_blocks.%d.0.conv.bias' % k])
        z = mlp(c, sd, u + 'inner_mlps.%d.0' % k)
        logits.append(z)
        probs.append(torch.sigmoid(z))
        if k == stages - 1:
            break
        prior = x_glob + make_block(occ[:, :k + 1], nbr, sd, u + 'outter_blocks.%d' % k)
    return logits, probs


def bits_of(probs, occ):
    """model_core.py:76-81: sum_k BCELoss(sum)(p_k, occ_k) / ln 2 (BCELoss clamps log at -100)."""
    bits = 0
    for k, p in enumerate(probs):
        bits = bits + F.binary_cross_entropy(p, occ[:, k:k + 1], reduction='sum') / math.log(2.0)
    return bits


def forward_scale(sd, scale):
    """LINR_PCGC_Model.forward for one scale dict {'offset_tensor','occ','nbr','scale_idx'} (torch tensors)."""
    x_low = scale_context(sd, scale['offset_tensor'], scale['scale_idx'])
    logits, probs = cnp_forward(sd, x_low, scale['occ'], scale['nbr'])
    return {'logits': logits, 'probs': probs, 'bits': bits_of(probs, scale['occ'])}


def frame_bits(sd, scales):
    """main.overfit_one_frame (main.py:457-475): sum of per-scale bits."""
    total = 0
    for s in scales:
        total = total + forward_scale(sd, s)['bits']
    return total


def to_torch_scales(np_scales, dtype=torch.float32):
    from . import octree
    out = []
    for s in np_scales:
        nbr = s['nbr'] if 'nbr' in s else octree.neighbour_table(s['coord'])
        out.append({'offset_tensor': torch.from_numpy(s['offset_tensor']).to(dtype),
                    'occ': torch.from_numpy(s['occ']).to(dtype),
                    'nbr': torch.from_numpy(nbr).long(), 'scale_idx': int(s['scale_idx'])})
    return out


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """torch.optim.Adam (L2 weight decay, no amsgrad) as configured at main.py:231-237; in-place on flat tensors."""
    g = grads + weight_decay * params
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    params.addcdiv_(exp_avg, denom, value=-lr / bc1)
