"""Parameter containers mirroring models/resnet.py (InceptionResNet, ResNetBlock) and ME.MinkowskiConvolution.

The arithmetic lives in the HIP engine (csrc/net.hip, csrc/spconv.hip); these modules own the parameters under the
reference's state-dict names (`.kernel` [27,Cin,Cout] or [Cin,Cout], `.bias` [1,Cout]) and initialise them the way
MinkowskiEngine 0.5.4 does: U(-1/sqrt(Cin*K), +1/sqrt(Cin*K)) for kernel and bias.
"""
import math

import torch
from torch import nn


class SparseConvolution(nn.Module):
    """Stand-in for ME.MinkowskiConvolution(in, out, kernel_size in {1,3}, stride=1, bias=True, dimension=3)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, bias=True, dimension=3):
        super().__init__()
        if kernel_size not in (1, 3) or stride != 1 or dimension != 3 or not bias:
            raise ValueError('only kernel_size in {1,3}, stride=1, bias=True, dimension=3 are on the hot path')
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        volume = kernel_size ** 3
        shape = (volume, in_channels, out_channels) if volume > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels))
        bound = 1.0 / math.sqrt(in_channels * volume)
        with torch.no_grad():
            self.kernel.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)


class InceptionResNet(nn.Module):
    """models/resnet.py:7-60: cat(conv3(relu(conv3(x))), conv1(relu(conv3(relu(conv1(x)))))) + x."""

    def __init__(self, channels, kernel_size=3, dimension=3):
        super().__init__()
        half = channels // 2
        self.conv0_0 = SparseConvolution(channels, half, kernel_size)
        self.conv0_1 = SparseConvolution(half, half, kernel_size)
        self.conv1_0 = SparseConvolution(channels, half, 1)
        self.conv1_1 = SparseConvolution(half, half, kernel_size)
        self.conv1_2 = SparseConvolution(half, half, 1)


class ResNetBlock(nn.Module):
    """models/resnet.py:146-162 with block_type='inception' (the only type any caller selects)."""

    def __init__(self, channels=32, kernel_size=3, block_layers=3, dimension=3, block_type='inception'):
        super().__init__()
        if block_type != 'inception':
            raise ValueError("only block_type='inception' is reachable from the reference's drivers")
        self.layers = nn.ModuleList([InceptionResNet(channels, kernel_size, dimension) for _ in range(block_layers)])
