#!/bin/bash
# Counter evidence of the non-spherical stress workload: the kernels of a training step on frame 0 of loot10 (sphere
# shell) and of loot10_rough (synthetic.rough_figure), both executors - HBM bytes per row (FETCH_SIZE / WRITE_SIZE, separate passes,
# FETCH doubled for gfx950), vector L1 hit rate (TCP -> TCC read requests per TCP access) and texture-addresser busy fraction, one counter
# group per rocprofv3 pass (kernel-trace only).   gpurun -- 'bash tools/rough_pmc.sh'   -> gpurun_out/rough_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/rough_pmc.txt
: > $OUT
for prec in f32 bf16; do
  for cfg in loot10 loot10_rough; do
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      d=/tmp/rp_${prec}_${cfg}_$i
      rm -rf $d
      if ! timeout -k 10 200 rocprofv3 --kernel-trace --kernel-include-regex "_k" --pmc $grp --output-format csv -d $d -- python3 $R/tools/traffic_probe.py 3 $prec $cfg > $d.log 2>&1; then
        echo "   pass failed or timed out ($prec $cfg: $grp): $(grep -m1 -i -E 'error|abort|fatal' $d.log) | $(tail -1 $d.log)" >> $OUT
        rm -rf $d
      fi
      echo "pass $prec $cfg $i done"
    done
  done
  python3 $R/tools/rough_pmc_summary.py $prec /tmp/rp_${prec}_loot10 /tmp/rp_${prec}_loot10_rough >> $OUT 2>&1
done
cat $OUT
