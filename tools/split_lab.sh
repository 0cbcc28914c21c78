#!/bin/bash
# Floors of the wave-specialised fused backward kernel: lab builds of csrc/fused_bwd.hip (-DFS_LAB=mask) timed with tools/fused_probe.py
# (nblocks 32: 32 blocks x ~55 tiles per pair, the executor's regime on an eighth of the chip).
#   on the build host:  bash tools/split_lab.sh build "0 1 2 4 8 32 ..."     on the GPU box: gpurun -- 'bash tools/split_lab.sh run'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/linr_pcgc_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $R/tools/_lab
  rm -f $R/tools/_lab/liblinr_fs_*.so
  for m in $2; do
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -DFS_LAB=$m $3 -c fused_bwd.hip -o $R/tools/_lab/fused_bwd_s$m.o &
  done
  wait
  for m in $2; do
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/liblinr_fs_$m.so $(ls _obj/*.o | grep -v '/fused_bwd.o$') $R/tools/_lab/fused_bwd_s$m.o -lpthread
  done
  ls $R/tools/_lab/liblinr_fs_*.so
else
  for f in $R/tools/_lab/liblinr_fs_*.so; do
    echo "$(basename $f): $(LINR_HIP_LIB=$f python3 $R/tools/fused_probe.py 20 ${2:-32} 2>&1 | grep fused)"
  done
  echo "single-stream: $(LINR_FUSED_SPLIT=0 python3 $R/tools/fused_probe.py 20 ${2:-32} 2>&1 | grep fused)"
fi
