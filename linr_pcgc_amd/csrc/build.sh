#!/bin/bash
# Builds liblinr_hip.so (gfx950 code objects + C-ABI) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../liblinr_hip.so
mkdir -p _obj
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall"
pids=()
for f in kmap spconv linear loss_optim net fused fused_bwd occ_wgrad net_bf16 train_bf16 decode octree wide; do
  if [ ! -f _obj/$f.o ] || [ $f.hip -nt _obj/$f.o ] || [ common.h -nt _obj/$f.o ] || [ conv_common.h -nt _obj/$f.o ] || [ layout.h -nt _obj/$f.o ] || [ bf16_common.h -nt _obj/$f.o ] || [ sce.h -nt _obj/$f.o ] || [ net_shared.h -nt _obj/$f.o ] || [ fused_bwd_split.h -nt _obj/$f.o ] || [ head_bwd.h -nt _obj/$f.o ] || [ ../../include/linr_hip.h -nt _obj/$f.o ]; then
    # fused_bwd / net_bf16: accumulators and destinations of the matrix instructions in VGPRs - fewer AGPR <-> VGPR copies in the
    # one-wave-per-SIMD kernels (same box: 1.6445 -> 1.629 ms/step; bf16 forward 0.436 -> 0.422 ms); no gain for the other files
    EXTRA=""
    if [ $f = fused_bwd ] || [ $f = net_bf16 ] || [ $f = train_bf16 ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
    if [ $f = occ_wgrad ]; then EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp"; fi          # -5 us/step (same box, twice); slower for fused.hip
    hipcc $FLAGS $EXTRA -c $f.hip -o _obj/$f.o &
    pids+=($!)
  fi
done
if [ ! -f _obj/ac.o ] || [ ac.cpp -nt _obj/ac.o ] || [ ../../include/linr_hip.h -nt _obj/ac.o ]; then
  g++ -O3 -fno-math-errno -fPIC -std=c++17 -fvisibility=hidden -Wall -c ac.cpp -o _obj/ac.o &
  pids+=($!)
fi
if [ ! -f _obj/ply.o ] || [ ply.cpp -nt _obj/ply.o ] || [ ../../include/linr_hip.h -nt _obj/ply.o ]; then
  g++ -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -c ply.cpp -o _obj/ply.o &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/kmap.o _obj/spconv.o _obj/linear.o _obj/loss_optim.o _obj/net.o _obj/fused.o _obj/fused_bwd.o _obj/occ_wgrad.o _obj/net_bf16.o _obj/train_bf16.o _obj/decode.o _obj/octree.o _obj/wide.o _obj/ac.o _obj/ply.o -lpthread
echo "built $(realpath $OUT)"
