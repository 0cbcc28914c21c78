#!/bin/bash
# Matrix-core utilisation of the conv-type kernels of a training step, executor configuration (tools/traffic_probe.py):
#   gpurun -- 'bash tools/mfma_pmc.sh'     (SQ counters only; one rocprofv3 --pmc pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_mfma.txt
: > $OUT
rm -rf /tmp/mf_1
timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "conv_bwd_wgrad_k|cconv_mfma_k|cconv_dual44_k|spconv_wgrad_t_k|occ_conv7_k|occ_wgrad7_k" --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/mf_1 -- python3 $R/tools/traffic_probe.py 3 > /tmp/mf_1.log 2>&1 || { tail -5 /tmp/mf_1.log; exit 1; }
python3 $R/tools/pmc_summary.py /tmp/mf_1 "conv_bwd_wgrad_k,cconv_mfma_k,cconv_dual44_k,spconv_wgrad_t_k,occ_conv7_k,occ_wgrad7_k" >> $OUT
cat $OUT
