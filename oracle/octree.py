"""Oracle (test infrastructure): numpy restatement of the per-frame octree prep.

Follows datautils/custom_dataset.py:259-355 (MyDataset.handle_data), models/module_utils.py:86-127
(octree_level.forward / upper_layer), :246-318 (QuickSearchCoord), models/quantize_functions.py:19-30
and models/sort_functions.py:46-60.  Pinned by tests/golden/octree_*.npz, which were produced by the
reference's own helpers (tests/golden/make_golden.py).
"""
import numpy as np

# glob_params.py:3 - 7-neighbour offsets (self, -x, +x, -y, +y, -z, +z)
OFFSETS_INI = np.array([[0, 0, 0], [-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]],
                       dtype=np.int64)
# module_utils.py:93 - child index i = 4*dx + 2*dy + dz
CHILD_OFFSETS = np.array([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], dtype=np.int64)


def ravel_key(c):
    """x-major key; monotone in the lexicographic (x, y, z) order for coords in [-1, 2^20)."""
    c = c.astype(np.int64) + 1
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def unique_sorted(c):
    """torch.unique(dim=0) == lexicographic sort + dedupe (custom_dataset.py:282, module_utils.py:248)."""
    c = np.asarray(c).astype(np.int64)
    key = ravel_key(c)
    _, idx = np.unique(key, return_index=True)
    return c[idx].astype(np.int32)


def contains(sorted_coords, query):
    """QuickSearchCoord.search with all-ones features (module_utils.py:260-275): 1.0 where present."""
    keys = ravel_key(sorted_coords)
    q = ravel_key(query)
    pos = np.searchsorted(keys, q)
    pos_c = np.minimum(pos, len(keys) - 1)
    return (pos < len(keys)) & (keys[pos_c] == q)   # queries are >= -1, so keys never go negative


def octree_level(coords):
    """module_utils.py:97-115: parent coords (unique floor(c/2)) and 8-column child occupancy."""
    coords = np.asarray(coords).astype(np.int64)
    parent = unique_sorted(coords // 2)
    occ = np.zeros((len(parent), 8), dtype=np.float32)
    for i in range(8):
        occ[:, i] = contains(coords, parent.astype(np.int64) * 2 + CHILD_OFFSETS[i])
    return parent, occ


def upper_layer(parent, occ):
    """module_utils.py:117-127: children of occupied octants, re-sorted x-major."""
    parent = np.asarray(parent).astype(np.int64) * 2
    kids = [parent[occ[:, i] == 1] + CHILD_OFFSETS[i] for i in range(8)]
    return unique_sorted(np.concatenate(kids, axis=0))


def offset_tensor(coords):
    """qscTensor.set_offset_tensor (module_utils.py:210-213): 7-neighbour occupancy, float {0,1}."""
    coords = np.asarray(coords).astype(np.int64)
    return np.stack([contains(coords, coords + o) for o in OFFSETS_INI], axis=1).astype(np.float32)


def prepare_frame(points, scale_num=None, min_point_num=64):
    """custom_dataset.py:259-355.  Returns the per-scale network inputs, finest scale first."""
    points = np.asarray(points)[:, :3].astype(np.int64)
    cmin = points.min(axis=0)
    xyz = unique_sorted(points - cmin)
    scales = []
    cur = xyz
    limit = 100000 if scale_num is None else scale_num
    for s in range(limit):
        parent, occ = octree_level(cur)
        scales.append({'coord': parent, 'occ': occ, 'offset_tensor': offset_tensor(parent),
                       'scale_idx': s, 'ground_truth': cur})
        if len(parent) < min_point_num or s == limit - 1:
            break
        cur = parent
    return {'scales': scales, 'point_num': int(len(xyz)), 'coord_data_min': cmin.astype(np.int32),
            'ori': xyz, 'scale_num': len(scales)}


def neighbour_table(coords):
    """Kernel map of a stride-1 3x3x3 sparse convolution on a fixed coordinate set.

    nbr[j, k] = row of coords[j] + delta_k or -1.  Offset enumeration follows MinkowskiEngine's hypercube
    kernel region with the first spatial axis fastest: k = (dx+1) + 3*(dy+1) + 9*(dz+1), centre k = 13
    (assumption - MinkowskiEngine source is not in the reference tree).
    """
    coords = np.asarray(coords).astype(np.int64)
    keys = ravel_key(coords)
    n = len(coords)
    nbr = np.full((n, 27), -1, dtype=np.int32)
    for k in range(27):
        d = np.array([k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1], dtype=np.int64)
        q = ravel_key(coords + d)
        pos = np.searchsorted(keys, q)
        pos_c = np.minimum(pos, n - 1)
        hit = (pos < n) & (keys[pos_c] == q)
        nbr[hit, k] = pos_c[hit]
    return nbr


def sphere_shell(bitdepth, radius, centre=None, thickness=0.5):
    """Synthetic voxelised sphere |‖p-c‖-r| < thickness, x-major sorted."""
    size = 1 << bitdepth
    c = np.array([size // 2] * 3 if centre is None else centre, dtype=np.int64)
    lo = np.maximum(c - int(radius + thickness) - 1, 0)
    hi = np.minimum(c + int(radius + thickness) + 2, size)
    out = []
    ys = np.arange(lo[1], hi[1])
    zs = np.arange(lo[2], hi[2])
    yy, zz = np.meshgrid(ys, zs, indexing='ij')
    d2yz = (yy - c[1]) ** 2 + (zz - c[2]) ** 2
    for x in range(lo[0], hi[0]):
        d = np.sqrt(d2yz + (x - c[0]) ** 2)
        m = np.abs(d - radius) < thickness
        if m.any():
            out.append(np.stack([np.full(m.sum(), x), yy[m], zz[m]], axis=1))
    return np.concatenate(out, axis=0).astype(np.int32)
