#!/bin/bash
# Two ranks of bench.py on ONE GPU (gloo rendezvous; RCCL refuses two ranks on one device): a rehearsal of the N > 1 control flow -
# headline per rank, the per-rank config[4] leg (8 frames instead of 64), the config[2] sequence split over the ranks - not a measurement.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LINR_BENCH_SINGLE_DEVICE=1 LINR_BENCH_BACKEND=gloo LINR_CONFIG4_FRAMES=${1:-8} LINR_SKIP_WIDE=1 LINR_SKIP_BPP_SEEDS=1 LINR_SKIP_ROUGH=1 \
  timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29549 \
  bench.py --gpus 2 --steps 20 --warmup 5 --seq-frames 96 --seq-epochs 4 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo rc=$?
python tools/bench_brief.py gpurun_out/bench_2rank.json | head -40
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_2rank.json').read().strip().splitlines()[-1])
print('config4_per_rank:', json.dumps(d.get('config4_per_rank')))
print('devices:', json.dumps(d.get('devices'))[:400])
PY
grep -v "amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM" gpurun_out/bench_2rank.err | tail -15
