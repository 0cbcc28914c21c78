#!/bin/bash
# Two ranks of the GOP-parallel sequence flow on ONE GPU (gloo rendezvous; RCCL refuses two ranks on one device): a rehearsal of the
# N > 1 control flow and of kernels running beside another process's, not a measurement.  args: frames gop epochs [schedule]
cd "$(dirname "$0")/.."
OUT=gpurun_out/rehearse_seq
rm -rf $OUT; mkdir -p $OUT
LINR_BENCH_SINGLE_DEVICE=1 LINR_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29547 -m linr_pcgc_amd.run --config loot10 --frames ${1:-96} --gop ${2:-32} \
  --first-epoch ${3:-4} --others-epoch ${3:-4} --schedule ${4:-static} --out $OUT --decode > $OUT/summary.json 2> $OUT/err.log
echo rc=$?
cat $OUT/summary.json
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/results_rank*.json')):
    for k,v in json.load(open(f)).items():
        print(f.split('/')[-1], k, v['gop'], 'rank', v['rank'], 'loss', [round(x,4) for x in v['loss']], 'bpp %.4f' % v['bpp']['bpp_all'], 'lossless', v['lossless'])
PY
grep -v "amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM" $OUT/err.log | tail -30
rm -rf $OUT/result_enc $OUT/output
