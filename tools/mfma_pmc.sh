#!/bin/bash
# Matrix-core utilisation of the two roofline kernels (executor configuration, tools/traffic_probe.py):
#   gpurun -- 'bash tools/mfma_pmc.sh'     (SQ counters only; one rocprofv3 --pmc pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_mfma.txt
: > $OUT
rm -rf /tmp/mf_1
timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "wgrad_t_k|wgrad_mfma|cconv_mfma" --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d /tmp/mf_1 -- python3 $R/tools/traffic_probe.py 5 > /tmp/mf_1.log 2>&1 || { tail -5 /tmp/mf_1.log; exit 1; }
python3 $R/tools/pmc_summary.py /tmp/mf_1 "wgrad_t_k,wgrad_mfma,cconv_mfma" >> $OUT
cat $OUT
