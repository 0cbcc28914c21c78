#!/bin/bash
# Lab builds of the library that differ only in the LLVM flags of csrc/fused_bwd.hip (the fp32 headline's dominant kernel):
#   bash tools/fb_flags_build.sh      (here, no GPU needed) -> tools/_lab/liblinr_fbflags_<n>.so + tools/_lab/fbflags.txt
# then on the GPU box:  bash tools/ab_lib.sh $(ls tools/_lab/liblinr_fbflags_*.so)
set -e
cd "$(dirname "$0")/../linr_pcgc_amd/csrc"
mkdir -p ../../tools/_lab
BASE="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall"
OBJS="_obj/kmap.o _obj/spconv.o _obj/linear.o _obj/loss_optim.o _obj/net.o _obj/fused.o _obj/occ_wgrad.o _obj/net_bf16.o _obj/train_bf16.o _obj/decode.o _obj/wide.o _obj/ac.o _obj/ply.o"
declare -a V=(
  "-mllvm -amdgpu-mfma-vgpr-form"
  ""
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-ilp"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-memory-clause"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-schedule-metric-bias=0"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-schedule-metric-bias=100"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-use-amdgpu-trackers=1"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -enable-post-misched=0"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -misched-bottomup"
  "-mllvm -amdgpu-mfma-vgpr-form -mllvm -misched-topdown"
)
: > ../../tools/_lab/fbflags.txt
for i in "${!V[@]}"; do
  if hipcc $BASE ${V[$i]} -c fused_bwd.hip -o /tmp/fb_$i.o 2>/tmp/fb_$i.err; then
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lab/liblinr_fbflags_$i.so $OBJS /tmp/fb_$i.o -lpthread
    echo "$i: ${V[$i]}" >> ../../tools/_lab/fbflags.txt
  else
    echo "$i: ${V[$i]}  -> DID NOT COMPILE: $(head -c 200 /tmp/fb_$i.err)" >> ../../tools/_lab/fbflags.txt
  fi
done
cat ../../tools/_lab/fbflags.txt
