#!/bin/bash
# SQ / LDS / TA counters of the bf16 training executor's kernels (one counter group per rocprofv3 pass, kernel-trace only):
#   gpurun -- 'bash tools/pmc_bf16.sh [steps]'   -> gpurun_out/pmc_bf16.txt   (per-launch averages)
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-12}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_bf16.txt
: > $OUT
i=0
KREG="bbwd_k|bconv_k|bocc|thead_bwd_k"
for grp in \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rm -rf /tmp/pb_$i
  echo "== pass $i: $grp" >> $OUT
  if timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "$KREG" --pmc $grp --output-format csv -d /tmp/pb_$i -- python3 $R/tools/bf16_steps.py bf16 $STEPS > /tmp/pb_$i.log 2>&1; then
    python3 $R/tools/pmc_summary.py /tmp/pb_$i "bbwd_k,bconv_k,bocc,thead_bwd_k" >> $OUT 2>&1 || echo "   (no counter file)" >> $OUT
  else
    echo "   pass failed or timed out: $(grep -m1 -i -E 'error|abort|fatal' /tmp/pb_$i.log) | $(tail -1 /tmp/pb_$i.log)" >> $OUT
  fi
done
cat $OUT
