#!/bin/bash
# Lab builds of the head backward (csrc/head_bwd.h: -DHB_LAB=mask, -DHB_GC_CHAINS=n) timed with tools/head_probe.py.
#   on the build host:  bash tools/head_lab.sh build "0 1 2 4 8 16 31" ["-DHB_GC_CHAINS=4"]    on the GPU box: gpurun -- 'bash tools/head_lab.sh run'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/linr_pcgc_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $R/tools/_lab
  tag=$(echo "$3" | tr -cd '0-9A-Za-z')
  for m in $2; do
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -DHB_LAB=$m $3 -c fused.hip -o $R/tools/_lab/fused_hb_$m$tag.o &
  done
  wait
  for m in $2; do
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/liblinr_hb_$m$tag.so $(ls _obj/*.o | grep -v '/fused.o$') $R/tools/_lab/fused_hb_$m$tag.o -lpthread
  done
  ls $R/tools/_lab/liblinr_hb_*.so
else
  echo "product: $(python3 $R/tools/head_probe.py 2>&1 | grep 'head bwd')"
  for f in $R/tools/_lab/liblinr_hb_*.so; do
    echo "$(basename $f): $(LINR_HIP_LIB=$f python3 $R/tools/head_probe.py 2>&1 | grep 'head bwd')"
  done
fi
