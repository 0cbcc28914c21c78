#!/bin/bash
# What the wide convolution kernels (csrc/wide.hip: wconv_k, wwgrad_k) wait for at hidden_channel_conv 16: SQ counters of their launches in
# 3 training steps, one counter group per rocprofv3 pass (kernel-trace only).   gpurun -- 'bash tools/wide_counters.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/wide_counters.txt
: > $OUT
i=0
KREG="wconv_k|wwgrad_k"
for grp in \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rm -rf /tmp/wdc_$i
  echo "== pass $i: $grp" >> $OUT
  if timeout -k 10 150 rocprofv3 --kernel-trace --kernel-include-regex "$KREG" --pmc $grp --output-format csv -d /tmp/wdc_$i -- python3 $R/tools/wide_step_prof.py 16 3 > /tmp/wdc_$i.log 2>&1; then
    python3 $R/tools/pmc_summary.py /tmp/wdc_$i "wconv_k,wwgrad_k" >> $OUT 2>&1 || echo "   (no counter file)" >> $OUT
  else
    echo "   pass failed or timed out: $(tail -2 /tmp/wdc_$i.log | tr '\n' ' ')" >> $OUT
  fi
done
cat $OUT
