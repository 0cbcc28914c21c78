#!/bin/bash
# BASELINE config[2] as a user runs it (nothing resident beforehand): 300 frames, GOP 32, 10 + 10 epochs on one GPU, with and without the
# decode of every frame and with / without background staging:  gpurun -- 'bash tools/config2_full.sh'  ->  gpurun_out/config2_full.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/config2_full.txt; : > $OUT
run() {
  rm -rf /tmp/c2; t0=$(date +%s.%N)
  timeout -k 10 300 python -m linr_pcgc_amd.run --config loot10 --frames 300 --gop 32 --first-epoch 10 --others-epoch 10 --out /tmp/c2 "$@" > /tmp/c2.json 2> /tmp/c2.err || { tail -3 /tmp/c2.err; return; }
  python3 -c "
import json; d=json.load(open('/tmp/c2.json')); print('%-48s wall %.2f s  %.4f s/frame  %.5f bits/point  lossless %s' % ('$*', d['wall_s'], d['sec_per_frame'], d['bits_per_point'], d['lossless']))" >> $OUT
}
run
run --no-stage-ahead
run --decode
run --decode --no-stage-ahead
cat $OUT
