#!/bin/bash
# HBM traffic of EVERY kernel of a training step (all of the library's kernels are named *_k), executor configuration (grouped launches): separate --pmc passes for FETCH_SIZE and
# WRITE_SIZE (KB per dispatch; FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes), averaged per kernel name.
#   gpurun -- 'bash tools/traffic_pmc.sh'   ->  gpurun_out/pmc_traffic.txt, gpurun_out/traffic.json  (copy the latter to profiles/)
#   gpurun -- 'bash tools/traffic_pmc.sh bf16' -> the same for the bf16 training executor: gpurun_out/pmc_traffic_bf16.txt, gpurun_out/traffic_bf16.json
PREC=${1:-f32}
SUF=""; [ "$PREC" != f32 ] && SUF="_$PREC"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_traffic$SUF.txt
: > $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "_k" --pmc $c --output-format csv -d /tmp/tr_$c -- python3 $R/tools/traffic_probe.py 3 $PREC > /tmp/tr_$c.log 2>&1 || { tail -5 /tmp/tr_$c.log; exit 1; }
  echo "pass $c done"
done
python3 $R/tools/traffic_summary.py /tmp/tr_FETCH_SIZE /tmp/tr_WRITE_SIZE $R/gpurun_out/traffic$SUF.json 3 "$(grep -h "^rows" /tmp/tr_FETCH_SIZE.log | tail -1)" $PREC | tee -a $OUT
