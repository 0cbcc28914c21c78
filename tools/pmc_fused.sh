#!/bin/bash
# SQ counters of the fused backward kernel (one counter group per pass):  gpurun -- 'bash tools/pmc_fused.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_fused.txt
: > $OUT
i=0
for grp in \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmcf_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "conv_bwd_wgrad_k|cconv_mfma_k" --pmc $grp --output-format csv -d /tmp/pmcf_$i -- python3 $R/tools/fused_probe.py 3 > /tmp/pmcf_$i.log 2>&1 || { tail -5 /tmp/pmcf_$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmcf_$i "conv_bwd_wgrad_k,cconv_mfma_k" >> $OUT
done
cat $OUT
