"""Data-set classes of the reference's drivers (datautils/custom_dataset.py) on this package's octree preparation.

``MyDataset`` lists the frames of a directory and turns a frame into the network's inputs - coordinates shifted to the origin and
de-duplicated, then per scale the parent voxels with their 8-column child occupancy and 7-neighbour occupancy
(custom_dataset.py:259-355) - through ``module_utils.prepare_frame`` (sorted-key searches; the occupancy from the kernel-map
builder's search kernel on a GPU).  Prepared frames are kept in RAM: the reference pickles every frame to ``handle_dir`` and
re-reads the pickle on every access of every epoch (custom_dataset.py:230-256); ``handle_dir`` is still created because the
drivers put their own files there (main.py:156-157) and delete it at the end (main.py:117-118).
``MytestDataset`` hands out the x-major sorted voxel list of a frame for the decoder's comparison (decoder.py:118-131).
"""
import os

import numpy as np
import torch

from . import ply
from .module_utils import OFFSETS_INI, prepare_frame, sort_by_coord_sum_c, unique_sorted

device = 'cuda' if torch.cuda.is_available() else 'cpu'


def read_ply_o3d(filedir, dtype='int32'):
    """custom_dataset.py:10-14 without open3d: the vertex coordinates of a PLY, rounded."""
    return ply.read_ply_xyz(filedir).astype(dtype)


def write_ply_ascii(filedir, coords, dtype='int32'):
    """custom_dataset.py:37-58: ASCII PLY with float x / y / z properties holding the integer coordinates."""
    coords = np.asarray(coords).astype(dtype).reshape(-1, 3)
    with open(filedir, 'w') as f:
        f.write('ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n' % coords.shape[0])
        np.savetxt(f, coords, fmt='%d')


def write_ply_o3d(filedir, coords, dtype='int32', normal=False, knn=None):
    """custom_dataset.py:16-35 (open3d's ASCII writer with the header patched to float): the same file write_ply_ascii makes.
    Normal estimation is open3d's and is not offered."""
    if normal:
        raise ValueError('normal estimation needs open3d')
    write_ply_ascii(filedir, coords, dtype)


class Read_Data:
    """custom_dataset.py:103-111: a window [idx_range[0], ...] onto a data set."""

    def __init__(self, dataset, idx_range):
        self.dataset, self.idx_range, self.offset_idx = dataset, idx_range, idx_range[0]

    def __getitem__(self, idx):
        return self.dataset[idx + self.offset_idx]

    def __len__(self):
        return len(self.idx_range)


class Read_Data_with_cache(Read_Data):
    """custom_dataset.py:113-119: touches every frame of the window once so that later accesses hit the cache."""

    def __init__(self, dataset, idx_range):
        super().__init__(dataset, idx_range)
        for idx in self.idx_range:
            self.dataset[idx]


def _list_frames(ori_dir, ori_type):
    names = sorted(n for n in os.listdir(ori_dir) if n.endswith('.' + ori_type) and not os.path.isdir(os.path.join(ori_dir, n)))
    if not names:
        raise ValueError('No file found in the directory')
    return [os.path.join(ori_dir, n) for n in names]


def _read(path, ori_type):
    if ori_type == 'npy':
        return np.asarray(np.load(path))[:, :3]
    if ori_type == 'ply':
        return ply.read_ply_xyz(path)
    raise ValueError('ori_type should be npy or ply')


class MytestDataset:
    """custom_dataset.py:123-152: frame idx as an int32 [P, 3] tensor sorted by the x-major key (sort_by_coord_sum_c)."""

    def __init__(self, ori_dir, ori_type='npy'):
        self.ori_type = ori_type
        self.all_files_path = _list_frames(ori_dir, ori_type)

    def __getitem__(self, idx):
        return self.handle_data(self.all_files_path[idx])

    def __len__(self):
        return len(self.all_files_path)

    def handle_data(self, file_path):
        pts = torch.as_tensor(np.asarray(_read(file_path, self.ori_type)).astype(np.int32), device=device)
        return sort_by_coord_sum_c(pts)          # sorted, NOT de-duplicated, like the reference


class MyDataset:
    """custom_dataset.py:155-355.  dataset[idx] -> {'all_input_info': per-scale dicts ('xyzqsc_t', 'ground_truth', 'scale_idx',
    'occ_lst'), 'xyzQ_low_bits', 'point_num', 'ori' (with derive_ori), 'coord_data_min'}; the first access fixes ``scale_num`` when
    it was None (main.py:77-78)."""

    _STAGES = {3: [[0, 7], [1, 6], [2, 3, 4, 5]], 4: [[0, 1], [2, 3], [4, 5], [6, 7]], 8: [[k] for k in range(8)]}

    def __init__(self, ori_dir, handle_dir=None, scale_num=None, ori_type='npy', stage=4, derive_neigbor=False, derive_ori=False):
        if stage not in self._STAGES:
            raise ValueError('stage must be 3, 4 or 8')
        if derive_neigbor:
            raise ValueError('derive_neigbor (neighbour index lists) is not used by any driver and is not offered')
        self.stage, self.stage_list = stage, self._STAGES[stage]
        self.ori_type, self.scale_num, self.derive_ori = ori_type, scale_num, derive_ori
        self.derive_neigbor = False
        self.load_cache = handle_dir is not None
        if handle_dir is not None:
            os.makedirs(handle_dir, exist_ok=True)
        self.all_files_path = _list_frames(ori_dir, ori_type)
        stem = lambda p: os.path.basename(p).split('.')[0]
        self.all_handle_path = [os.path.join(handle_dir, stem(p) + '.pkl') for p in self.all_files_path] if self.load_cache else []
        self.min_point_num = 64
        self.offsets_ini = OFFSETS_INI
        self.offset_of_neigbor = OFFSETS_INI
        self._ram = {}

    def __len__(self):
        return len(self.all_files_path)

    def split(self, data):
        return [data[:, cols] for cols in self.stage_list]

    def set_prefix_data(self, setdata):
        """custom_dataset.py:208-228.  The 7-neighbour offsets are the fixed ones of glob_params.py:3 (what the coding network's
        scale context is built for); another list is refused rather than silently ignored."""
        offs = setdata.get('offsets_ini')
        if offs is not None:
            got = torch.as_tensor(offs).reshape(-1, 3).cpu().to(torch.int64).tolist()
            if got != [list(o) for o in OFFSETS_INI]:
                raise ValueError('offsets_ini must be the 7 face-neighbour offsets of glob_params.py:3')
        self.min_point_num = setdata.get('min_point_num', 64)
        self._ram = {}

    def __getitem__(self, idx):
        if idx not in self._ram:
            self._ram[idx] = self.handle_data(self.all_files_path[idx])
        return self._ram[idx]

    def handle_data(self, file_path):
        fr = prepare_frame(_read(file_path, self.ori_type), self.scale_num, self.min_point_num, device=device, with_offsets=True)
        if self.scale_num is None:
            self.scale_num = fr['scale_num']
        for info in fr['all_input_info']:
            occ = info['occ']
            info['occ_lst'] = self.split(occ)
        low = fr['all_input_info'][-1]['coord']
        bitdepth_q = int(np.ceil(np.log2(int(low.max()) + 1)))
        cells = (2 ** bitdepth_q) ** 3
        n_low = int(low.shape[0])
        return {'all_input_info': fr['all_input_info'], 'xyzQ_low_bits': min(n_low, cells - n_low) * bitdepth_q * 3,
                'point_num': fr['point_num'], 'ori': fr['ori'] if self.derive_ori else None, 'coord_data_min': fr['coord_data_min']}


__all__ = ['MyDataset', 'MytestDataset', 'Read_Data', 'Read_Data_with_cache', 'read_ply_o3d', 'write_ply_ascii', 'write_ply_o3d',
           'OFFSETS_INI', 'unique_sorted']
