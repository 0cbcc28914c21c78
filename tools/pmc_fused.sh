#!/bin/bash
# SQ / TA counters of the fp32 executor's convolution kernels inside 3 training steps (one counter group per pass):  gpurun -- 'bash tools/pmc_fused.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_fused.txt
: > $OUT
i=0
for grp in \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAVES" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcf_$i
  echo "== pass $i: $grp" >> $OUT
  if timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "conv_bwd_wgrad_k|cconv_mfma_k" --pmc $grp --output-format csv -d /tmp/pmcf_$i -- python3 $R/tools/traffic_probe.py 3 > /tmp/pmcf_$i.log 2>&1; then
    python3 $R/tools/pmc_summary.py /tmp/pmcf_$i "conv_bwd_wgrad_k,cconv_mfma_k" >> $OUT
  else
    echo "   pass failed or timed out: $(grep -m1 -i -E 'error|abort|fatal|exceed|cannot' /tmp/pmcf_$i.log) | last: $(tail -1 /tmp/pmcf_$i.log)" >> $OUT
  fi
done
cat $OUT
