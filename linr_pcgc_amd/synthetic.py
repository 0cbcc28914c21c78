"""Synthetic voxelised point-cloud sequences standing in for 8iVFB / MVUB / Owlii (no datasets in the image).

SURVEY.md §8d: config 1 = 8-bit sphere r=100; config 2/3 = 10-bit sphere r~250 with an integer, seed-free per-frame
jitter (loot stand-in, ~784 k points); config 4 = 2-voxel-thick 10-bit shell r=228 (andrew10 stand-in).
"""
import math

import numpy as np


def sphere_shell(bitdepth, radius, centre=None, thickness=0.5):
    """Voxels with | ||p - c|| - r | < thickness, x-major sorted, int32 [P,3]."""
    size = 1 << bitdepth
    c = np.array([size // 2] * 3 if centre is None else centre, dtype=np.int64)
    ext = int(math.ceil(radius + thickness)) + 1
    lo, hi = np.maximum(c - ext, 0), np.minimum(c + ext + 1, size)
    ys, zs = np.arange(lo[1], hi[1]), np.arange(lo[2], hi[2])
    yy, zz = np.meshgrid(ys, zs, indexing='ij')
    d2 = (yy - c[1]) ** 2 + (zz - c[2]) ** 2
    r_in, r_out = max(radius - thickness, 0.0), radius + thickness
    chunks = []
    for x in range(lo[0], hi[0]):
        d = d2 + (x - c[0]) ** 2
        m = (d > r_in * r_in) & (d < r_out * r_out)
        if m.any():
            chunks.append(np.stack([np.full(int(m.sum()), x, dtype=np.int64), yy[m], zz[m]], axis=1))
    return np.concatenate(chunks, axis=0).astype(np.int32)


CONFIGS = {
    'sphere8': {'bitdepth': 8, 'radius': 100, 'thickness': 0.5},       # BASELINE config 1 (125,810 points)
    'loot10': {'bitdepth': 10, 'radius': 250, 'thickness': 0.5},       # BASELINE config 2/3 (784,314 points)
    'andrew10': {'bitdepth': 10, 'radius': 228, 'thickness': 1.0},     # BASELINE config 4 (1,306,322 points)
    'owlii11': {'bitdepth': 11, 'radius': 480, 'thickness': 0.5},      # BASELINE config 5
}


def sphere_shell_device(bitdepth, radius, centre, thickness, device, slab=32):
    """The same voxel list as sphere_shell, enumerated on the GPU (x slabs of the bounding box; integer distances compared
    with the float64 thresholds, so the result is identical).  Returns an int32 [P,3] tensor on `device`, x-major sorted.
    A 784 k-point frame takes ~0.7 s in sphere_shell's numpy loop and a few milliseconds here - what makes the 300-frame
    sequence of BASELINE config[2] practical to stage."""
    import torch
    size = 1 << bitdepth
    c = [int(v) for v in centre]
    ext = int(math.ceil(radius + thickness)) + 1
    lo = [max(v - ext, 0) for v in c]
    hi = [min(v + ext + 1, size) for v in c]
    r_in, r_out = max(radius - thickness, 0.0), radius + thickness
    t_in, t_out = r_in * r_in, r_out * r_out
    ys = torch.arange(lo[1], hi[1], device=device, dtype=torch.int64)
    zs = torch.arange(lo[2], hi[2], device=device, dtype=torch.int64)
    d2 = ((ys - c[1]) ** 2)[:, None] + ((zs - c[2]) ** 2)[None, :]
    chunks = []
    for x0 in range(lo[0], hi[0], slab):
        xs = torch.arange(x0, min(x0 + slab, hi[0]), device=device, dtype=torch.int64)
        d = (d2[None, :, :] + ((xs - c[0]) ** 2)[:, None, None]).to(torch.float64)
        idx = ((d > t_in) & (d < t_out)).nonzero()                  # row-major = x, then y, then z: the x-major order
        if idx.shape[0]:
            chunks.append(torch.stack([xs[idx[:, 0]], ys[idx[:, 1]], zs[idx[:, 2]]], dim=1).to(torch.int32))
    if not chunks:
        return torch.zeros((0, 3), dtype=torch.int32, device=device)
    return torch.cat(chunks, dim=0)


def sequence_params(config, t):
    cfg = CONFIGS[config] if isinstance(config, str) else config
    size = 1 << cfg['bitdepth']
    cx = size // 2 + int(math.floor(3 * math.sin(2 * math.pi * t / 30)))
    cz = size // 2 + int(math.floor(2 * math.cos(2 * math.pi * t / 45)))
    r = cfg['radius'] + int(math.floor(4 * math.sin(2 * math.pi * t / 20)))
    return cfg['bitdepth'], r, (cx, size // 2, cz), cfg['thickness']


def sequence_frame_device(config, t, device='cuda'):
    """sequence_frame(config, t) enumerated on `device` (identical points, int32 tensor)."""
    b, r, c, th = sequence_params(config, t)
    return sphere_shell_device(b, r, c, th, device)


def sequence_frame(config, t):
    """Frame t of the synthetic sequence: centre += (floor(3 sin(2 pi t/30)), 0, floor(2 cos(2 pi t/45))),
    radius += floor(4 sin(2 pi t/20)) - integer, seed-free motion."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    size = 1 << cfg['bitdepth']
    cx = size // 2 + int(math.floor(3 * math.sin(2 * math.pi * t / 30)))
    cz = size // 2 + int(math.floor(2 * math.cos(2 * math.pi * t / 45)))
    r = cfg['radius'] + int(math.floor(4 * math.sin(2 * math.pi * t / 20)))
    return sphere_shell(cfg['bitdepth'], r, (cx, size // 2, cz), cfg['thickness'])
