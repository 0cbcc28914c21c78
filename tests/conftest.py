import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle issues tens of thousands of small torch ops; on a 256-core GPU host each of them would fan out over
    # every core (measured 10x slower than 8 threads)
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(8, n)))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='module')
def pkg():
    """The package with its HIP library loaded (GPU tests): raises if liblinr_hip.so is missing - there is no fallback."""
    import linr_pcgc_amd  # noqa: F401
    from linr_pcgc_amd import _lib
    _lib.lib()
    return linr_pcgc_amd


@pytest.fixture(scope='module')
def shell(golden_dir):
    """The golden 128-cube shell (reference-generated octree fixture) with the oracle's neighbour tables."""
    import numpy as np
    from oracle import octree as ooct
    g = np.load(os.path.join(golden_dir, 'octree_shell128.npz'))
    scales = []
    for s in range(int(g['scale_num'])):
        c = g['s%d_coord' % s]
        scales.append({'coord': c, 'occ': g['s%d_occ' % s], 'offset_tensor': g['s%d_offset' % s], 'scale_idx': s,
                       'nbr': ooct.neighbour_table(c)})
    return {'scales': scales, 'point_num': int(len(g['ori']))}
