"""Host-side mirror of models/module_utils.py: PointwiseMLP, BinaryArithmeticCoding, octree helpers.

PointwiseMLP owns parameters only (its arithmetic runs in csrc/linear.hip through the engine).
BinaryArithmeticCoding feeds the C++ range coder of csrc/ac.cpp (streams follow torchac 0.9.3's published coder).
The octree helpers restate octree_level / QuickSearchCoord (models/module_utils.py:86-318) with 64-bit ravel keys and
torch.searchsorted; they are device-agnostic torch code and run once per frame ("next" row N1 of SURVEY.md §8f).
"""
import os

import numpy as np
import torch
from torch import nn

from . import _lib


class PointwiseMLP(nn.Sequential):
    """models/module_utils.py:42-81: Linear(+ReLU) stack, xavier_uniform(gain=sqrt(2)) weights, zero bias."""

    def __init__(self, dims, doLastRelu=False, init_bias='zeros'):
        layers = []
        for i in range(1, len(dims)):
            fc = nn.Linear(dims[i - 1], dims[i])
            nn.init.xavier_uniform_(fc.weight, gain=nn.init.calculate_gain('relu'))
            if init_bias == 'uniform':
                nn.init.uniform_(fc.bias)
            elif init_bias == 'zeros':
                nn.init.zeros_(fc.bias)
            else:
                raise ValueError('Unknown init ' + str(init_bias))
            layers.append(fc)
            if i < len(dims) - 1 or doLastRelu:
                layers.append(nn.ReLU())
        super().__init__(*layers)


class BinaryArithmeticCoding:
    """models/module_utils.py:8-40 on top of linr_ac_encode_binary / linr_ac_decode_binary (cdf = [0, 1-p, 1])."""

    @staticmethod
    def _host(prob):
        return np.ascontiguousarray(torch.as_tensor(prob).detach().reshape(-1).to('cpu', torch.float32).numpy())

    def encode(self, prob, occupancy):
        p = self._host(prob)
        s = np.ascontiguousarray(torch.as_tensor(occupancy).detach().reshape(-1).to('cpu').numpy().astype(np.uint8))
        if p.shape != s.shape:
            raise ValueError('prob and occupancy must have the same number of elements')
        cap = 2 * p.size + 64                              # worst case is 16 bit per symbol
        out = np.empty(cap, dtype=np.uint8)
        n = _lib.lib().linr_ac_encode_binary(p.ctypes.data, s.ctypes.data, p.size, out.ctypes.data, cap)
        if n < 0:
            _lib.check(int(n), 'linr_ac_encode_binary')
        return out[:n].tobytes()

    def decode(self, prob, bitstream):
        p = self._host(prob)
        buf = np.frombuffer(bitstream, dtype=np.uint8)
        out = np.empty(p.size, dtype=np.uint8)
        _lib.check(_lib.lib().linr_ac_decode_binary(p.ctypes.data, p.size, buf.ctypes.data if buf.size else None,
                                                    buf.size, out.ctypes.data), 'linr_ac_decode_binary')
        return torch.from_numpy(out.astype(np.int16))

    def estimate_bitrate(self, prob, occupancy):
        p = torch.as_tensor(prob).reshape(-1).double()
        t = torch.as_tensor(occupancy).reshape(-1).double()
        return float(-(torch.log2(torch.where(t > 0.5, p, 1 - p))).sum())


# ---- octree / neighbourhood prep -------------------------------------------------------------------------------------
OFFSETS_INI = ((0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1))     # glob_params.py:3


def ravel_key(coord):
    """x-major key for coordinates in [-1, 2^20); equal ordering to sort_by_coor_sum_detail (sort_functions.py:46-60)."""
    c = coord.to(torch.int64) + 1
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def unique_sorted(coord, coord_bits=None):
    """torch.unique(dim=0): de-duplicated rows in x-major order, int32.  On the GPU one library call (key sort + unique:
    linr_coords_sort_unique) for coordinates in [0, 2^20); torch.unique otherwise.  coord_bits: the caller vouches that every
    coordinate is in [0, 2^coord_bits) (no range check, fewer radix passes)."""
    if coord.is_cuda and coord.shape[0] > 0 and coord.dtype in (torch.int32, torch.int64):
        from . import ops
        if coord_bits is not None:
            return ops.coords_sort_unique(coord.to(torch.int32).contiguous(), 0, coord_bits)
        lo, hi = torch.aminmax(coord)
        if int(lo) >= 0 and int(hi) < (1 << 20):
            return ops.coords_sort_unique(coord.to(torch.int32).contiguous(), 0)
    key = torch.unique(ravel_key(coord))
    mask = (1 << 21) - 1
    return torch.stack([(key >> 42) - 1, ((key >> 21) & mask) - 1, (key & mask) - 1], dim=1).to(torch.int32)


def sort_by_coord_sum_c(xyz, onlycoord=True):
    """models/sort_functions.py:17-30: the rows in x-major order, duplicates kept (what the decoder compares against, decoder.py:10)."""
    if not onlycoord:
        raise ValueError('only the coordinate form is used by the drivers')
    lo = xyz.min() - 1
    c = xyz.to(torch.int64) - lo.to(torch.int64)
    return xyz[torch.argsort((c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2], stable=True)]


def contains(sorted_coord, query):
    """QuickSearchCoord.search with all-ones features (module_utils.py:260-275): float {0,1} column."""
    keys = ravel_key(sorted_coord)
    q = ravel_key(query)
    pos = torch.searchsorted(keys, q)
    pos_c = pos.clamp(max=keys.numel() - 1)
    return ((pos < keys.numel()) & (keys[pos_c] == q)).to(torch.float32)


class octree_level(nn.Module):
    """module_utils.py:86-127: parent coordinates + 8-column child occupancy, and its inverse."""

    def __init__(self):
        super().__init__()
        self.offsets = torch.tensor([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], dtype=torch.int64)

    def forward(self, leaf, qsc=None, coord_bits=None):
        """coord_bits: the caller vouches that every coordinate is in [0, 2^coord_bits) (prepare_frame does); None: checked here -
        the library's keys hold 20 bits per axis and its kernels do not validate (negative or wider values would build garbage
        keys silently), so anything outside takes the torch path below, which handles the full int range like the reference."""
        if leaf.is_cuda and leaf.dtype == torch.int32 and leaf.numel() > 0:
            if coord_bits is None:
                lo, hi = torch.aminmax(leaf)
                coord_bits = 20 if int(lo) >= 0 and int(hi) < (1 << 20) else 0
            if 0 < coord_bits <= 20:      # GPU: parents (key sort + unique of leaf >> 1) and their child occupancy (sorted-key search)
                from . import ops         # in ONE library call
                return ops.octree_level(leaf.contiguous(), coord_bits)
        parent = unique_sorted(torch.div(leaf.to(torch.int64), 2, rounding_mode='floor'))
        off = self.offsets.to(leaf.device)
        # all 8 child lookups of every parent as ONE batched search over the sorted leaf keys
        q = (parent.to(torch.int64)[:, None, :] * 2 + off[None, :, :]).reshape(-1, 3)
        occ = contains(leaf, q).reshape(-1, 8)
        return parent, occ

    def upper_layer(self, parent_C, occupancy):
        off = self.offsets.to(parent_C.device)
        base = parent_C.to(torch.int64) * 2
        kids = torch.cat([base[occupancy[:, i] == 1] + off[i] for i in range(8)], dim=0)
        return unique_sorted(kids)


octree_level_obj = octree_level()


class qscTensor:
    """module_utils.py:155-224 (the members the drivers use): sorted unique coords + 7-neighbour occupancy."""

    def __init__(self, coord, feat=None, presorted=False, coord_bits=None):
        self.coord_bits = coord_bits                  # set by prepare_frame: every coordinate is in [0, 2^coord_bits)
        self.coord = coord if presorted else unique_sorted(coord, coord_bits)
        self.feat = feat
        self.parent_C = self.occupancy = self.offset_tensor = None

    def get_coord(self):
        return self.coord

    def search(self, coord_in):
        return contains(self.coord, coord_in).reshape(-1, 1)

    def set_oct_level(self):
        if self.coord_bits is not None:
            self.parent_C, self.occupancy = octree_level_obj(self.coord, coord_bits=self.coord_bits)
        else:
            self.parent_C, self.occupancy = octree_level_obj(self.coord)

    def get_oct_level(self):
        return self.parent_C, self.occupancy

    def upper_layer(self, parent, occupancy):
        return octree_level_obj.upper_layer(parent, occupancy)

    def set_offset_tensor(self, offsets=OFFSETS_INI):
        offs = torch.as_tensor(offsets, dtype=torch.int64, device=self.coord.device)
        c = self.coord.to(torch.int64)
        q = (c[:, None, :] + offs[None, :, :]).reshape(-1, 3)
        self.offset_tensor = contains(self.coord, q).reshape(-1, offs.shape[0])

    def get_offset_tensor(self):
        return self.offset_tensor


def prepare_frame(points, scale_num=None, min_point_num=64, device='cpu', with_offsets=True):
    """MyDataset.handle_data (datautils/custom_dataset.py:259-355) for an in-memory point list [P,3].
    with_offsets=False (the GOP drivers' lean form) leaves 'offset_tensor' None: engine.Frame then reads the 7-neighbour occupancy off
    the kernel map it builds anyway (linr_kmap_offset_feat) and overfit.Gop stores those rows back into the dicts; it also leaves
    'occ_lst' (the eight [N,1] column views of 'occ' that the reference's model.forward takes) None - the executor takes 'occ' as one
    matrix, LINR_PCGC_Model makes the views itself when it is handed such a dict, and 56 view objects per frame were 6 % of a frame's
    staging time."""
    dev = torch.device(device)
    if torch.is_tensor(points):                    # device-resident input (synthetic.sequence_frame_device): no host round trip
        pts = points[:, :3].to(device=dev)
    else:
        pts = torch.as_tensor(np.asarray(points)[:, :3], device=dev)
    if pts.shape[0] == 0:
        raise ValueError('the frame has no points')
    if dev.type == 'cuda' and not pts.dtype.is_floating_point and (pts.dtype == torch.int32 or int(pts.abs().max()) < (1 << 30)):
        # GPU: minimum / span by one kernel, the shift by coord_data_min inside the key kernel of the sort (torch's int64 column
        # reductions alone cost 0.64 ms per 784 k-point frame)
        from . import ops
        p32 = pts.to(torch.int32).contiguous()
        mm = ops.coords_minmax(p32)
        mmh = mm.tolist()
        cmin = torch.tensor(mmh[:3], dtype=torch.int64)
        span = max(mmh[3 + a] - mmh[a] for a in range(3))
        if span >= (1 << 20):
            raise ValueError('the cloud spans %d voxels along an axis; the kernel map holds 20-bit coordinates (the data sets of the '
                             'reference are 10 to 12 bit)' % (span + 1))
        bits = max(1, int(span).bit_length())
        cur = qscTensor(ops.coords_sort_unique(p32, 0, bits, origin=mm[:3]), presorted=True, coord_bits=bits)
    else:
        pts = pts.to(torch.int64)
        cmin = pts.min(dim=0).values
        span = int((pts.max(dim=0).values - cmin).max())
        if span >= (1 << 20):
            raise ValueError('the cloud spans %d voxels along an axis; the kernel map holds 20-bit coordinates (the data sets of the '
                             'reference are 10 to 12 bit)' % (span + 1))
        bits = max(1, int(span).bit_length())             # every shifted coordinate is in [0, 2^bits); one bit less per level
        cur = qscTensor(pts - cmin, coord_bits=bits)
    ori = cur.get_coord()
    info = []
    limit = 100000 if scale_num is None else scale_num
    s0 = 0
    if dev.type == 'cuda' and not pts.dtype.is_floating_point and 2 <= bits <= 11 and not os.environ.get('LINR_OCTREE_PER_LEVEL'):
        # every level in one library call and one host read (csrc/octree.hip: no sort, counts chained on the device); the per-level
        # loop below only continues where this leaves off (it does not: the stop criterion is met inside unless scale_num asks for
        # levels whose coordinates have no bits left)
        from . import ops
        lv = ops.octree_levels(ori, bits, min(limit, 64))
        if lv is not None:
            par_all, occ_all, counts = lv
            off, child = 0, ori
            for s0, n in enumerate(counts):
                parent, occ = par_all[off:off + n], occ_all[off:off + n]
                off += n
                bits = max(1, bits - 1)
                low = qscTensor(parent, presorted=True, coord_bits=bits)
                if with_offsets:
                    low.set_offset_tensor()
                info.append({'xyzqsc_t': low, 'coord': low.get_coord(), 'offset_tensor': low.get_offset_tensor(),
                             'ground_truth': child, 'scale_idx': s0, 'occ': occ, 'occ_lst': list(occ.split(1, dim=1)) if with_offsets else None})
                child = parent
                if n < min_point_num or s0 == limit - 1:
                    return {'all_input_info': info, 'point_num': int(ori.shape[0]), 'ori': ori,
                            'coord_data_min': cmin.to('cpu', torch.int32).tolist(), 'scale_num': len(info)}
            cur = low
            s0 = len(counts)
    for s in range(s0, limit):
        cur.set_oct_level()
        parent, occ = cur.get_oct_level()
        bits = max(1, bits - 1)
        low = qscTensor(parent, presorted=True, coord_bits=bits)
        if with_offsets:
            low.set_offset_tensor()
        info.append({'xyzqsc_t': low, 'coord': low.get_coord(), 'offset_tensor': low.get_offset_tensor(),
                     'ground_truth': cur.get_coord(), 'scale_idx': s, 'occ': occ,
                     'occ_lst': list(occ.split(1, dim=1)) if with_offsets else None})
        if parent.shape[0] < min_point_num or s == limit - 1:
            break
        cur = low
    return {'all_input_info': info, 'point_num': int(ori.shape[0]), 'ori': ori,
            'coord_data_min': cmin.to('cpu', torch.int32).tolist(), 'scale_num': len(info)}
