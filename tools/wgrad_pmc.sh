#!/bin/bash
# SQ counters of the weight-gradient kernel variants (single launches on frame 0 of loot10):  gpurun -- 'bash tools/wgrad_pmc.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_wgrad_variants.txt
: > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  rm -rf /tmp/wv_1
  timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "wgrad" --pmc $grp --output-format csv -d /tmp/wv_1 -- python3 $R/tools/wgrad_variants_probe.py 5 > /tmp/wv_1.log 2>&1 || { tail -5 /tmp/wv_1.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/wv_1 "wgrad" >> $OUT
done
cat $OUT
