#!/bin/bash
# HBM traffic of the two roofline kernels: separate --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_traffic.txt
: > $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "wgrad_t_k|wgrad_mfma|cconv_mfma" --pmc $c --output-format csv -d /tmp/tr_$c -- python3 $R/tools/traffic_probe.py 5 > /tmp/tr_$c.log 2>&1 || { tail -5 /tmp/tr_$c.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/tr_$c "wgrad_t_k,wgrad_mfma,cconv_mfma" >> $OUT
  echo "pass $c done"
done
cat $OUT
