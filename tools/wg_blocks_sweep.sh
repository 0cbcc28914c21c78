#!/bin/bash
# LINR_WG_BLOCKS sweep on the GPU box (gpurun -- 'bash tools/wg_blocks_sweep.sh'): rebuild + default bench per value.
# Measured (ms/step): 512: 2.704, 256: 2.708, 192: 2.848, 128: 3.027, 64: 3.577 (768: 2.790 on a ~3 % slower box).
R=$GRAFT_REPO_ROOT
for nb in ${NBS:-512 256 192 128 64}; do
  sed -i "s/^#define LINR_WG_BLOCKS [0-9]*/#define LINR_WG_BLOCKS $nb/" $R/linr_pcgc_amd/csrc/common.h
  (cd $R && bash linr_pcgc_amd/csrc/build.sh > /dev/null 2>&1)
  echo "WG_BLOCKS $nb: $(cd $R && LINR_SKIP_ROOFLINE=1 python bench.py --no-cpu-baseline 2>/dev/null | cut -c100-150)"
done
