#!/bin/bash
# Builds the WG_LAB variants of the 8->8 weight-gradient kernel used by tools/wgrad_floor_lab.sh into tools/_lab/ (git-ignored;
# the .so files travel to the GPU box).  The variants are a patch against fused.hip of commit a34a1a3:
#   0 as shipped then | 1 no gathers | 2 gathers + 1 of 8 MFMAs | 3 arithmetic indices (no index / gradient loads) | 4 MFMAs + loop
#   only | 5 no cin_valid branches | 6 gathers issued in consumption order | 7 = 5 + 6 | 8 saddr-form gathers | 9 = 6 + 8 (adopted)
#   | 10 = 9 without branches | 11 / 12 s_setprio around gathers (with / without branches) | 13 inverse priority | 14 table holds
#   index + 1 | 15 = 14 + 11 | 16 idle lanes' gathers masked off
# usage: bash tools/wgrad_floor_lab_build.sh "0 1 2 ..."      (run linr_pcgc_amd/csrc/build.sh first: the other objects are linked as built)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_lab
git show a34a1a3:linr_pcgc_amd/csrc/fused.hip > tools/_lab/fused_lab.hip
patch -s tools/_lab/fused_lab.hip tools/wgrad_floor_lab.patch
O=linr_pcgc_amd/csrc/_obj
for v in ${1:-0 1 2 3 4 5}; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -DWG_LAB=$v -Ilinr_pcgc_amd/csrc -c tools/_lab/fused_lab.hip -o tools/_lab/fused_lab$v.o &
done
wait
for v in ${1:-0 1 2 3 4 5}; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_lab/liblinr_lab$v.so $O/kmap.o $O/spconv.o $O/linear.o $O/loss_optim.o $O/net.o tools/_lab/fused_lab$v.o $O/net_bf16.o $O/ac.o -lpthread
done
ls tools/_lab/*.so
