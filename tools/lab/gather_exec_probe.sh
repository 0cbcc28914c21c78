#!/bin/bash
# EXEC-masked neighbour gathers against the zero-row read, on the REAL kernel map of a loot10 frame (tools/lab/gather_exec_probe.hip).
# Run on the GPU box from the repo root:  bash tools/lab/gather_exec_probe.sh > gpurun_out/gather_exec.txt
set -e
mkdir -p tools/_lab gpurun_out
hipcc --offload-arch=gfx950 -O3 -o tools/_lab/gather_exec_probe tools/lab/gather_exec_probe.hip
python3 - <<'PY'
import torch
from linr_pcgc_amd import overfit, synthetic
gop = overfit.Gop(None, [synthetic.sequence_frame_device('loot10', 0, 'cuda')], None, 64, 'cuda')
f = gop.frames[0]
f.nbr.cpu().numpy().tofile('/tmp/nbr_loot10.bin')
open('/tmp/nbr_loot10.rows', 'w').write('%d' % f.rows)
print('dumped', tuple(f.nbr.shape), f.rows)
PY
./tools/_lab/gather_exec_probe /tmp/nbr_loot10.bin $(cat /tmp/nbr_loot10.rows)
./tools/_lab/gather_exec_probe
