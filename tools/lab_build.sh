#!/bin/bash
# Lab build of the library with some csrc files recompiled under extra flags:  bash tools/lab_build.sh <file[,file..] without .hip> <tag> "<flags>"
#   -> tools/_lab/liblinr_<tag>.so (time it with LINR_HIP_LIB=<that file>, e.g. tools/ab_bf16.sh / tools/ab_lib.sh); the kernels' registers: tools/_lab/<file>_<tag>.res
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/linr_pcgc_amd/csrc
files=$(echo $1 | tr ',' ' '); tag=$2; shift 2
mkdir -p $R/tools/_lab
objs=$(ls _obj/*.o)
for f in $files; do
  EXTRA=""
  if [ $f = fused_bwd ] || [ $f = net_bf16 ] || [ $f = train_bf16 ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
  if [ $f = occ_wgrad ]; then EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden $EXTRA "$@" -Rpass-analysis=kernel-resource-usage -c $f.hip -o $R/tools/_lab/${f}_$tag.o 2> $R/tools/_lab/${f}_$tag.res || { tail -20 $R/tools/_lab/${f}_$tag.res; exit 1; }
  objs="$(echo "$objs" | grep -v "/$f.o$") $R/tools/_lab/${f}_$tag.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/liblinr_$tag.so $objs -lpthread
echo "$R/tools/_lab/liblinr_$tag.so"
