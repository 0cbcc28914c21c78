"""Point-cloud file input of the data set (datautils/custom_dataset.py:9-14,263-269: `read_ply_o3d` / `np.load`):
x, y, z of a PLY (ascii, binary_little_endian or binary_big_endian; any extra vertex properties are skipped) or of an
.npy array, as an int64 [P, 3] array.  open3d is not needed."""
import numpy as np

_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2', 'ushort': 'u2',
              'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4',
              'double': 'f8', 'float64': 'f8'}


def read_ply_xyz(path):
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError('%s is not a PLY file' % path)
        fmt, n_vertex, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError('%s: PLY header is not terminated' % path)
            tok = line.decode('ascii', 'replace').split()
            if not tok or tok[0] == 'comment':
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    n_vertex = int(tok[2])
                elif n_vertex is None:
                    raise ValueError('%s: an element precedes the vertex element' % path)
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError('%s: list property in the vertex element' % path)
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        if fmt is None or n_vertex is None:
            raise ValueError('%s: no format / vertex element in the header' % path)
        names = [p[0] for p in props]
        if not all(k in names for k in ('x', 'y', 'z')):
            raise ValueError('%s: vertex element has no x / y / z' % path)
        if fmt == 'ascii':
            # the sequences of the data sets are ASCII: one pass of the library's host-side parser over the text (0.7 s -> tens of
            # milliseconds for a loot frame; the C call does not hold the GIL, so read_many scales over frames)
            import ctypes
            from . import _lib
            text = f.read()
            out = np.empty((n_vertex, 3), dtype=np.int64)
            done = ctypes.c_int64(0)
            cols = [names.index(k) for k in ('x', 'y', 'z')]
            rc = _lib.lib().linr_ply_parse_ascii(text, len(text), n_vertex, len(names), cols[0], cols[1], cols[2],
                                                 out.ctypes.data, ctypes.byref(done))
            if rc != 0:
                raise ValueError('%s: malformed vertex line %d (of %d announced)' % (path, done.value + 1, n_vertex))
            return out
        else:
            endian = '<' if fmt == 'binary_little_endian' else '>'
            dt = np.dtype([(n, endian + t) for n, t in props])
            rec = np.frombuffer(f.read(n_vertex * dt.itemsize), dtype=dt, count=n_vertex)
            xyz = np.stack([rec['x'], rec['y'], rec['z']], axis=1)
    if xyz.shape[0] != n_vertex:
        raise ValueError('%s: %d vertices announced, %d read' % (path, n_vertex, xyz.shape[0]))
    return np.rint(xyz).astype(np.int64)


def read_many(paths, workers=None):
    """The frames of a GOP, read and parsed on a thread pool (file reads and the C parser both release the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    paths = list(paths)
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    workers = max(1, min(int(workers), 16, len(paths)))
    if workers == 1:
        return [read_points(p) for p in paths]
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(read_points, paths))


def read_points(path):
    """`ori_type` 'ply' or 'npy' of the reference's data set, chosen by the file extension."""
    if str(path).lower().endswith('.npy'):
        return np.asarray(np.load(path))[:, :3].astype(np.int64)
    return read_ply_xyz(path)


def write_ply_xyz(path, xyz, binary=True):
    """Minimal writer (tests, exporting decoded frames): float x y z like open3d's write_point_cloud."""
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    header = 'ply\nformat %s 1.0\ncomment linr_pcgc_amd\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n' \
        % ('binary_little_endian' if binary else 'ascii', xyz.shape[0])
    with open(path, 'wb') as f:
        f.write(header.encode('ascii'))
        if binary:
            f.write(xyz.astype('<f4').tobytes())
        else:
            np.savetxt(f, xyz, fmt='%g')
