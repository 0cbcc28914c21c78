#!/bin/bash
# Floors of the fused backward kernel: lab builds of csrc/fused_bwd.hip (-DFB_LAB=mask) timed with tools/fused_probe.py.
#   on the build host:  bash tools/fused_lab.sh build "0 16 1 2 4 8 3 ..."     on the GPU box: gpurun -- 'bash tools/fused_lab.sh run'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/linr_pcgc_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $R/tools/_lab
  for m in $2; do
    # the product's flags for this file (csrc/build.sh) and EVERY other object of the library, whatever build.sh links today
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -DFB_LAB=$m -c fused_bwd.hip -o $R/tools/_lab/fused_bwd_$m.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/liblinr_fb_$m.so $(ls _obj/*.o | grep -v '/fused_bwd.o$') $R/tools/_lab/fused_bwd_$m.o -lpthread
  done
  ls $R/tools/_lab/*.so
else
  for f in $R/tools/_lab/liblinr_fb_*.so; do
    echo "$(basename $f): $(LINR_HIP_LIB=$f python3 $R/tools/fused_probe.py 20 ${2:-256} 2>&1 | grep fused)"
  done
fi
