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
elif [ "$1" = prof ]; then
  # per-kernel times of every lab build (KIND 0 / 1 / 2 apart): rocprofv3 --kernel-trace --stats over the probe
  cd /tmp && export TMPDIR=/tmp
  for f in $R/tools/_lab/liblinr_fs_*.so; do
    rm -rf /tmp/sl_prof
    LINR_HIP_LIB=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sl_prof -- python3 $R/tools/fused_probe.py 20 ${2:-32} > /tmp/sl_prof.log 2>&1
    echo "$(basename $f): $(python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/sl_prof/*/*kernel_stats.csv')
rows = list(csv.DictReader(open(f[0]))) if f else []
print('  '.join('%s %.1f' % (r['Name'].split('(')[0].replace('void conv_bwd_wgrad_k', 'K'), float(r['AverageNs']) / 1e3) for r in sorted(rows, key=lambda r: r['Name']) if 'conv_bwd_wgrad_k' in r['Name']))
PY
)"
  done
else
  for f in $R/tools/_lab/liblinr_fs_*.so; do
    echo "$(basename $f): $(LINR_HIP_LIB=$f python3 $R/tools/fused_probe.py 20 ${2:-32} 2>&1 | grep fused)"
  done
  echo "single-stream: $(LINR_FUSED_SPLIT=0 python3 $R/tools/fused_probe.py 20 ${2:-32} 2>&1 | grep fused)"
fi
