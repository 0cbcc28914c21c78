"""Host-side arithmetic coder speed (ns / binary symbol) on this machine."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (loads libamdhip64 before the library)
from linr_pcgc_amd import _lib  # noqa: E402

L = _lib.lib()
rng = np.random.default_rng(0)
n = 2_000_000
p = np.clip(rng.beta(0.3, 0.3, size=n), 1e-4, 1 - 1e-4).astype(np.float32)
s = (rng.random(n) < p).astype(np.uint8)
out = np.empty(2 * n + 64, np.uint8)
dec = np.empty(n, np.uint8)
for rep in range(3):
    t = time.time(); ln = L.linr_ac_encode_binary(p.ctypes.data, s.ctypes.data, n, out.ctypes.data, out.size); te = time.time() - t
    t = time.time(); L.linr_ac_decode_binary(p.ctypes.data, n, out.ctypes.data, ln, dec.ctypes.data); td = time.time() - t
    print('bits/sym %.3f  enc %.1f ns/sym  dec %.1f ns/sym  ok=%s' % (ln * 8 / n, te / n * 1e9, td / n * 1e9, bool((dec == s).all())))
print('cpus', len(os.sched_getaffinity(0)))
