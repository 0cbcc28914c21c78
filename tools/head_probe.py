"""Times the fp32 head backward (linr_head_bwd: csrc/head_bwd.h) on 8 x the rows of a loot10 frame in ONE group - the tiles per
wave of the executor's grouped launch over the 8 heads.  usage: head_probe.py [reps]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import _lib
L = _lib.lib()
dev = 'cuda'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 8 * 336529
g = torch.Generator(device=dev).manual_seed(1)
c = torch.randn(n, 8, device=dev, generator=g)
p = torch.rand(n, device=dev, generator=g) * 0.98 + 0.01
t = (torch.rand(n, 8, device=dev, generator=g) < 0.5).float()
w1 = torch.randn(24, 8, device=dev, generator=g) * 0.3
b1 = torch.randn(24, device=dev, generator=g) * 0.1
w2 = torch.randn(24, device=dev, generator=g) * 0.3
gc = torch.empty(n, 8, device=dev)
gh = torch.empty(241, device=dev)
ws = torch.empty(L.linr_head_workspace_bytes(n), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    _lib.check(L.linr_head_bwd(c.data_ptr(), p.data_ptr(), t.data_ptr(), 8, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), 1.0, gc.data_ptr(), n,
                               gh.data_ptr(), ws.data_ptr(), ws.numel(), st), 'linr_head_bwd')
for _ in range(50): run()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / reps * 1e3)
print('head bwd + slab reduce, %d rows: %.1f us  (gc sum %.6g, ghead sum %.6g)' % (n, best, float(gc.double().sum()), float(gh.double().sum())))
