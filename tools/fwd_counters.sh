#!/bin/bash
# What saturates in the forward convolution cconv_mfma_k<8,8,fwd>: SQ / LDS / vector-memory / L2 counters of
# the executor's own launches in a training step, one counter group per rocprofv3 pass (kernel-trace only).  TA_* / TCP_* groups
# have hung rocprofv3 on this pool before (tools/README.md): they run LAST, each under its own short timeout, and a pass that
# fails is reported and skipped.   gpurun -- 'bash tools/fwd_counters.sh'   -> gpurun_out/fwd_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/fwd_counters.txt
: > $OUT
i=0
KREG="cconv_mfma_k|cconv_dual44_k|conv_bwd_wgrad_k"
for grp in \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "TCC_READ_sum TCC_WRITE_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/fwc_$i
  echo "== pass $i: $grp" >> $OUT
  if timeout -k 10 150 rocprofv3 --kernel-trace --kernel-include-regex "$KREG" --pmc $grp --output-format csv -d /tmp/fwc_$i -- python3 $R/tools/traffic_probe.py 3 > /tmp/fwc_$i.log 2>&1; then
    python3 $R/tools/pmc_summary.py /tmp/fwc_$i "cconv_mfma_k,cconv_dual44_k,conv_bwd_wgrad_k" >> $OUT 2>&1 || echo "   (no counter file)" >> $OUT
  else
    # the FIRST error line says why (r04: five TCC counters in one pass aborted rocprofv3 - the TCC block has 4 slots per pass)
    echo "   pass failed or timed out: $(grep -m1 -i -E 'error|abort|fatal|exceed|cannot' /tmp/fwc_$i.log) | last: $(tail -1 /tmp/fwc_$i.log)" >> $OUT
  fi
done
cat $OUT
