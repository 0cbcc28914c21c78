"""Per-frame staging cost split (generator / octree levels / kernel maps), synchronous, loot10 frames 0..7 - what bench.py's
`staging_ms_per_frame` reports.   python tools/stage_split.py [config]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import overfit          # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'loot10'
for rep in range(2):
    print(overfit.staging_split(cfg, range(8 * rep, 8 * rep + 8), 'cuda'))
