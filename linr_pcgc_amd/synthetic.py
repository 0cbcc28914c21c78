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


def sphere_shell_device(bitdepth, radius, centre, thickness, device, slab=192):
    """The same voxel list as sphere_shell, enumerated on the GPU.  Per (x, y) column only the z CANDIDATES around the two roots
    +-sqrt(r^2 - dx^2 - dy^2) are tested (two windows of at most W voxels, W ~ 3 away from the equator and ~ sqrt(2 r thickness) at
    it) - 13 M tests for a 784 k-point frame where the whole bounding box has 125 M - with the SAME exact criterion: integer squared
    distances against the float64 thresholds, so the result is identical to sphere_shell's (the float sqrt only places the windows,
    padded by a voxel on both sides).  Returns an int32 [P,3] tensor on `device`, x-major sorted.  ~0.3 ms per 784 k-point frame (the
    dense enumeration took 3 ms, numpy 0.7 s): what makes the cold 300-frame sequence of BASELINE config[2] a staging measurement
    rather than a generator measurement."""
    import torch
    size = 1 << bitdepth
    c = [int(v) for v in centre]
    ext = int(math.ceil(radius + thickness)) + 1
    lo = [max(v - ext, 0) for v in c]
    hi = [min(v + ext + 1, size) for v in c]
    r_in, r_out = max(radius - thickness, 0.0), radius + thickness
    t_in, t_out = r_in * r_in, r_out * r_out
    ys = torch.arange(lo[1], hi[1], device=device, dtype=torch.int64)
    dy2 = (ys - c[1]) ** 2
    # window width: zhi - zlo <= sqrt(t_out - d2) - sqrt(t_in - d2) + 3 <= sqrt(t_out - t_in) + 3 (a host constant: no device read)
    W = int(math.sqrt(t_out - t_in)) + 6
    chunks = []
    for x0 in range(lo[0], hi[0], slab):
        xs = torch.arange(x0, min(x0 + slab, hi[0]), device=device, dtype=torch.int64)
        d2 = ((xs - c[0]) ** 2)[:, None] + dy2[None, :]                                  # [nx, ny] int64
        d2f = d2.to(torch.float64)
        zhi = torch.sqrt((t_out - d2f).clamp(min=0.0)).ceil().to(torch.int64) + 1       # |dz| candidates: zlo - 1 .. zhi + 1 (padded)
        zlo = torch.sqrt((t_in - d2f).clamp(min=0.0)).floor().to(torch.int64) - 1
        live = d2f < t_out
        p0 = torch.maximum(zlo, 1 - zlo)                                                 # first dz of the positive window (disjoint from the negative one)
        j = torch.arange(W, device=device, dtype=torch.int64)
        # side 0: dz = -zhi + j (ascending z) while dz <= -zlo;  side 1: dz = p0 + j while dz <= zhi
        dz0 = -zhi[:, :, None] + j
        dz1 = p0[:, :, None] + j
        ok0 = live[:, :, None] & (dz0 <= -zlo[:, :, None]) & (dz0 < p0[:, :, None])
        ok1 = live[:, :, None] & (dz1 <= zhi[:, :, None])
        dz = torch.stack([dz0, dz1], dim=2)                                              # [nx, ny, 2, W]
        ok = torch.stack([ok0, ok1], dim=2)
        d = (d2[:, :, None, None] + dz * dz).to(torch.float64)
        z = c[2] + dz
        m = ok & (d > t_in) & (d < t_out) & (z >= lo[2]) & (z < hi[2])
        idx = m.nonzero()                                                                 # row-major = x, y, side, j = x, y, z ascending
        if idx.shape[0]:
            chunks.append(torch.stack([xs[idx[:, 0]], ys[idx[:, 1]], z[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]]], dim=1).to(torch.int32))
    if not chunks:
        return torch.zeros((0, 3), dtype=torch.int32, device=device)
    return torch.cat(chunks, dim=0)


def describe(config):
    """One phrase on a config's geometry (bench.py's workload string)."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    if cfg.get('figure'):
        return '%d-bit rough figure: generalised cylinders with thin parts, +-8 voxel displacement' % cfg['bitdepth']
    return '%d-bit sphere shell r~%d, %g voxel thick' % (cfg['bitdepth'], cfg['radius'], 2 * cfg['thickness'])


def sequence_params(config, t):
    cfg = CONFIGS[config] if isinstance(config, str) else config
    size = 1 << cfg['bitdepth']
    cx = size // 2 + int(math.floor(3 * math.sin(2 * math.pi * t / 30)))
    cz = size // 2 + int(math.floor(2 * math.cos(2 * math.pi * t / 45)))
    r = cfg['radius'] + int(math.floor(4 * math.sin(2 * math.pi * t / 20)))
    return cfg['bitdepth'], r, (cx, size // 2, cz), cfg['thickness']


def sequence_frame_device(config, t, device='cuda'):
    """sequence_frame(config, t) enumerated on `device` (identical points, int32 tensor)."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    if cfg.get('figure'):
        return rough_figure_device(cfg['bitdepth'], t, device)
    b, r, c, th = sequence_params(config, t)
    return sphere_shell_device(b, r, c, th, device)


def sequence_frame(config, t):
    """Frame t of the synthetic sequence: centre += (floor(3 sin(2 pi t/30)), 0, floor(2 cos(2 pi t/45))),
    radius += floor(4 sin(2 pi t/20)) - integer, seed-free motion."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    if cfg.get('figure'):
        return rough_figure(cfg['bitdepth'], t)
    size = 1 << cfg['bitdepth']
    cx = size // 2 + int(math.floor(3 * math.sin(2 * math.pi * t / 30)))
    cz = size // 2 + int(math.floor(2 * math.cos(2 * math.pi * t / 45)))
    r = cfg['radius'] + int(math.floor(4 * math.sin(2 * math.pi * t / 20)))
    return sphere_shell(cfg['bitdepth'], r, (cx, size // 2, cz), cfg['thickness'])


# ---- a non-spherical stress workload: 'loot10_rough' ------------------------------------------------------------------------------
# Every stand-in above is a sphere shell - the smoothest surface there is - while the reference's data are scanned human figures
# (datautils/custom_dataset.py:259-355 on 8iVFB PLYs, loot/info.log:3-26).  The rough figure is a union of generalised cylinders along
# y - torso, head, two legs, two thin slanted arms and a thin sheet - each given by per-y TABLES of its elliptical cross-section
# (centre cx, cz and semi-axes ax, az in 1/8 voxel), inflated by a low-frequency displacement of +-8 voxels
#   d(x, y, z) = (SX[x] * SY[y] * SZ[z]) >> 12   (tables in [-64, 64], periods 97 / 131 / 113 voxels),
# voxelised as the boundary of the solid (an inside voxel with a 6-neighbour outside).  The tables are built on the host in float64
# and rounded to integers; everything per voxel is int64 arithmetic, so the numpy and the GPU enumeration give identical points.
# Motion is integer and seed-free like the sphere sequences: the figure translates, the arms swing, the wrinkles drift.
# Never a default, never the headline: bench.py's `rough` leg and tests/test_gpu_configs.py use it beside the spheres.
def _figure_tables(bitdepth, t):
    size = 1 << bitdepth
    s = size / 1024.0
    y = np.arange(size, dtype=np.float64)
    tx = int(math.floor(3 * math.sin(2 * math.pi * t / 30)))
    tz = int(math.floor(2 * math.cos(2 * math.pi * t / 45)))
    swing = math.sin(2 * math.pi * t / 25)

    def ellipsoid(cx, cy, cz, rx, ry, rz):
        u = (y - cy * s) / (ry * s)
        prof = np.sqrt(np.clip(1.0 - u * u, 0.0, None))
        return np.full(size, cx * s), np.full(size, cz * s), rx * s * prof, rz * s * prof

    def limb(x0, y0, z0, x1, y1, z1, r0, r1, flat=1.0):
        """tapered, slanted cylinder from (x0, y0, z0) to (x1, y1, z1) (y0 > y1) with rounded ends; flat < 1 squeezes it in z"""
        u = np.clip((y0 * s - y) / ((y0 - y1) * s), 0.0, 1.0)
        cx, cz = (x0 + (x1 - x0) * u) * s, (z0 + (z1 - z0) * u) * s
        r = (r0 + (r1 - r0) * u) * s
        cap_hi = np.clip((y - y0 * s) / (r0 * s), 0.0, 1.0)
        cap_lo = np.clip((y1 * s - y) / (r1 * s), 0.0, 1.0)
        prof = np.sqrt(np.clip(1.0 - cap_hi ** 2, 0.0, None)) * np.sqrt(np.clip(1.0 - cap_lo ** 2, 0.0, None))
        return cx, cz, r * prof, r * prof * flat

    parts = [ellipsoid(512, 560, 512, 118, 200, 78),                                  # torso
             ellipsoid(512, 836, 522, 64, 80, 68),                                    # head
             limb(468, 420, 510, 446, 70, 524, 52, 30),                               # legs
             limb(556, 420, 510, 584, 70, 496, 52, 30),
             limb(404, 700, 512, 318 - 10 * swing, 410, 548 + 40 * swing, 26, 15),    # thin, slanted, swinging arms
             limb(620, 700, 512, 712 + 10 * swing, 430, 476 - 40 * swing, 26, 15),
             limb(512, 760, 584, 512, 330, 640 + 12 * swing, 150, 120, flat=0.03)]    # a thin sheet behind the torso
    tabs = []
    for cx, cz, ax, az in parts:
        ax8, az8 = np.rint(8 * ax).astype(np.int64), np.rint(8 * az).astype(np.int64)
        ax8[az8 <= 0] = 0
        tabs.append((np.rint(cx).astype(np.int64) + tx, np.rint(cz).astype(np.int64) + tz, ax8, az8))
    i = np.arange(size, dtype=np.float64)
    sx = np.rint(64 * np.sin(2 * math.pi * (i + 2 * t) / (97 * s))).astype(np.int64)
    sy = np.rint(64 * np.sin(2 * math.pi * i / (131 * s) + 0.7)).astype(np.int64)
    sz = np.rint(64 * np.cos(2 * math.pi * (i - t) / (113 * s))).astype(np.int64)
    live = np.zeros(size, dtype=bool)
    xlo, xhi, zlo, zhi = size, 0, size, 0
    for cx, cz, ax8, az8 in tabs:
        on = ax8 > 0
        live |= on
        if on.any():
            xlo = min(xlo, int((cx[on] - (ax8[on] + 71) // 8 - 2).min()))
            xhi = max(xhi, int((cx[on] + (ax8[on] + 71) // 8 + 3).max()))
            zlo = min(zlo, int((cz[on] - (az8[on] + 71) // 8 - 2).min()))
            zhi = max(zhi, int((cz[on] + (az8[on] + 71) // 8 + 3).max()))
    ys = np.nonzero(live)[0]
    box = (max(xlo, 1), min(xhi, size - 1), max(int(ys.min()) - 1, 1), min(int(ys.max()) + 2, size - 1), max(zlo, 1), min(zhi, size - 1))
    return tabs, (sx, sy, sz), box


def _figure_points(bitdepth, t, xp, to_dev, slab):
    """Shared enumeration: xp = numpy or torch; to_dev moves a host int64 array to where xp computes."""
    tabs, (sx, sy, sz), (x0, x1, y0, y1, z0, z1) = _figure_tables(bitdepth, t)
    ya, za = slice(y0 - 1, y1 + 1), slice(z0 - 1, z1 + 1)                       # one voxel of halo: the box never touches the cube's faces
    sy_d, sz_d = to_dev(sy[ya])[None, :, None], to_dev(sz[za])[None, None, :]
    zc = to_dev(np.arange(z0 - 1, z1 + 1, dtype=np.int64))[None, None, :]
    part_d = [(to_dev(cx[ya])[None, :, None], to_dev(cz[ya])[None, :, None],
               to_dev(ax8[ya])[None, :, None], to_dev(az8[ya])[None, :, None]) for cx, cz, ax8, az8 in tabs]

    def inside(xa, xb):
        xc = to_dev(np.arange(xa, xb, dtype=np.int64))[:, None, None]
        d = (to_dev(sx[xa:xb])[:, None, None] * sy_d * sz_d) >> 12             # 1/8 voxel, [-64, 64]
        acc = None
        for cx, cz, ax8, az8 in part_d:
            AX, AZ = ax8 + d, az8 + d
            DX, DZ = 8 * (xc - cx), 8 * (zc - cz)
            m = (ax8 > 0) & (AX > 0) & (AZ > 0) & ((DX * AZ) ** 2 + (DZ * AX) ** 2 < (AX * AZ) ** 2)
            acc = m if acc is None else (acc | m)
        return acc

    chunks = []
    for xa in range(x0, x1, slab):
        xb = min(xa + slab, x1)
        v = inside(xa - 1, xb + 1)                                              # [xb - xa + 2, ny + 2, nz + 2]
        c = v[1:-1, 1:-1, 1:-1]
        full = v[:-2, 1:-1, 1:-1] & v[2:, 1:-1, 1:-1] & v[1:-1, :-2, 1:-1] & v[1:-1, 2:, 1:-1] & v[1:-1, 1:-1, :-2] & v[1:-1, 1:-1, 2:]
        idx = (c & ~full).nonzero()                                             # row-major = x, then y, then z: the x-major order
        if xp is np:
            if len(idx[0]):
                chunks.append(np.stack([idx[0] + xa, idx[1] + y0, idx[2] + z0], axis=1).astype(np.int32))
        elif idx.shape[0]:
            import torch
            off = torch.tensor([xa, y0, z0], dtype=torch.int64, device=idx.device)
            chunks.append((idx + off).to(torch.int32))
    return chunks


def rough_figure(bitdepth=10, t=0, slab=16):
    """Frame t of the rough figure at `bitdepth` (numpy; ~10 s at 10 bit - tests use 7 or 8 bit)."""
    chunks = _figure_points(bitdepth, t, np, lambda a: a, slab)
    return np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 3), np.int32)


def rough_figure_device(bitdepth=10, t=0, device='cuda', slab=16):
    """rough_figure(bitdepth, t) enumerated on `device` (identical points, int32 tensor, x-major sorted)."""
    import torch
    chunks = _figure_points(bitdepth, t, torch, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device), slab)
    return torch.cat(chunks, dim=0) if chunks else torch.zeros((0, 3), dtype=torch.int32, device=device)


CONFIGS['loot10_rough'] = {'bitdepth': 10, 'figure': True}      # ~0.8 M points; the stress workload beside loot10
CONFIGS['rough8'] = {'bitdepth': 8, 'figure': True}             # its small sibling for the oracle-parity tests
