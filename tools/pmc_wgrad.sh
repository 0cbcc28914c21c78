#!/bin/bash
# PMC passes over tools/wgrad_lab.py (one counter group per run):  gpurun -- 'bash tools/pmc_wgrad.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_wgrad.txt
: > $OUT
i=0
for grp in \
  "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "wgrad|cconv_mfma" --pmc $grp --output-format csv -d /tmp/pmc_$i -- python3 $R/tools/wgrad_lab.py 3 > /tmp/pmc_$i.log 2>&1 || { tail -5 /tmp/pmc_$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_$i "wgrad,cconv_mfma" >> $OUT
  echo "pass $i done $(date +%T)" | tee -a $R/gpurun_out/pmc_progress.txt
done
cat $OUT
