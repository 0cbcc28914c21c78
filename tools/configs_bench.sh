#!/bin/bash
# ms/step, encode s/frame and bits/point of the other BASELINE configs' stand-ins on the final build (GOP of 8 frames, the complete
# 10-epoch overfit):  gpurun -- 'bash tools/configs_bench.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in sphere8 loot10 andrew10 owlii11; do
  LINR_SKIP_ROOFLINE=1 LINR_SKIP_BPP_SEEDS=1 LINR_SKIP_WIDE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-sequence --config $cfg --gop 8 --steps 20 --warmup 5 2>/tmp/cfg_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1])
    c=d['components_s_per_frame']
    print('%-9s %s' % ('$cfg', d['config']['workload'].split(': ',1)[1].split(', 1 GOP')[0].strip()))
    print('          %.3f ms/step  encode %.4f s/frame (overfit %.4f + codec %.4f)  %.4f bits/point  decode single %.4f s  lossless %s  bf16 codec fwd %.3f vs %.3f ms' % (d['ms_per_step'], d['value'], c['overfit'], c['codec_modelcomp_fwd_ac_write'], d['bits_per_point'], c['decode_s_single_frame'], d['lossless_decode_frames0to3'], d['bf16_codec']['forward_ms_per_frame']['bf16'], d['bf16_codec']['forward_ms_per_frame']['f32']))
except Exception as e:
    print('$cfg FAILED', repr(e)); print(open('/tmp/cfg_err.txt').read()[-600:])"
done
