"""Parameter containers mirroring models/upsample.py: ConvWithPrune and CNP (child-node predictor).

Registration order (= state-dict / parameters() order the model codec relies on): block_in, inner_mlps, inner_blocks
(empty at instage=1), prune_blocks, outter_blocks (models/upsample.py:43-76).  CNP.forward/encode/decode of the
reference are executed by the HIP engine through LINR_PCGC_Model (model_core.py in this package).
"""
from torch import nn

from .module_utils import PointwiseMLP
from .resnet import ResNetBlock, SparseConvolution


class ConvWithPrune(nn.Module):
    """models/upsample.py:13-23: a 3x3x3 convolution evaluated on the prior's own coordinates."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, dimension=3):
        super().__init__()
        self.conv = SparseConvolution(in_channels, out_channels, kernel_size, stride, bias, dimension)


def get_stage_lst(stage):
    """models/upsample.py:24-35."""
    table = {8: [[0], [1], [2], [3], [4], [5], [6], [7]], 4: [[0, 1], [2, 3], [4, 5], [6, 7]],
             3: [[0, 1], [6, 7], [2, 3, 4, 5]], 2: [[0, 1, 6, 7], [2, 3, 4, 5]], 1: [[0, 1, 2, 3, 4, 5, 6, 7]]}
    return table[stage]


class CNP(nn.Module):
    def __init__(self, in_channels=1, channels=12, kernel_size=3, block_layers=2, outstage=8, instage=1):
        super().__init__()
        if outstage != 8 or instage != 1:
            raise ValueError('the reference drivers only build outstage=8, instage=1 (main.py:97,218)')
        self.outstage, self.instage = outstage, instage
        self.block_in = self.make_block(in_channels, channels, channels, kernel_size, block_layers)
        widths = [len(s) for s in get_stage_lst(outstage)]
        self.inner_mlps = nn.ModuleList(
            [nn.ModuleList([PointwiseMLP([channels, 24, widths[k]]) for _ in range(instage)]) for k in range(outstage)])
        self.inner_blocks = nn.ModuleList([nn.ModuleList([]) for _ in range(outstage)])
        self.prune_blocks = nn.ModuleList(
            [nn.ModuleList([ConvWithPrune(channels, channels, 3) for _ in range(instage)]) for _ in range(outstage)])
        cum = 0
        outter = []
        for k in range(outstage - 1):
            cum += widths[k]
            outter.append(self.make_block(cum, channels, channels, kernel_size, 1))
        self.outter_blocks = nn.ModuleList(outter)

    @staticmethod
    def make_block(in_channels=32, channels=32, out_channels=32, kernel_size=3, block_layers=3):
        """models/upsample.py:88-97: conv3 -> ReLU -> ResNetBlock -> conv3."""
        return nn.Sequential(SparseConvolution(in_channels, channels, kernel_size), nn.ReLU(inplace=True),
                             ResNetBlock(channels, kernel_size, block_layers),
                             SparseConvolution(channels, out_channels, kernel_size))
