#!/bin/bash
# SQ counters of the block-local convolution probe beside the lane = row baseline:  gpurun -- 'bash tools/lab/lconv_pmc.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/lconv_pmc.txt
: > $OUT
i=0
for grp in \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf /tmp/lcp_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "base_k|lconv4_k<0>|lconv3_k<0>" --pmc $grp --output-format csv -d /tmp/lcp_$i -- $R/tools/_lab/lconv_probe 512 > /tmp/lcp_$i.log 2>&1 || { tail -5 /tmp/lcp_$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/lcp_$i "base_k,lconv4_k,lconv3_k" >> $OUT
done
cat $OUT
