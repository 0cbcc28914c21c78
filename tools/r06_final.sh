#!/bin/bash
# The round's closing evidence on the library that ships, one GPU session:
#   gpurun --timeout 1200 -- 'bash tools/r06_final.sh'     -> gpurun_out/r06_final/*   (copy what is to be judged into profiles/r06_final_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
D=$R/gpurun_out/r06_final
rm -rf $D; mkdir -p $D
echo "library: $(ls -l --time-style=+%Y-%m-%dT%H:%M:%S linr_pcgc_amd/liblinr_hip.so)" > $D/README.txt
# 1. counter bytes of every kernel of a step, both executors (separate --pmc passes), then into profiles/ for the bench line
bash tools/traffic_pmc.sh > $D/traffic_f32.log 2>&1 && cp gpurun_out/traffic.json $D/traffic.json && cp gpurun_out/traffic.json profiles/traffic.json
bash tools/traffic_pmc.sh bf16 > $D/traffic_bf16.log 2>&1 && cp gpurun_out/traffic_bf16.json $D/traffic_bf16.json && cp gpurun_out/traffic_bf16.json profiles/traffic_bf16.json
echo "traffic done" >> $D/README.txt
# 2. SQ / TA counters of the dominant kernels, both executors
bash tools/pmc_fused.sh > /dev/null 2>&1; cp gpurun_out/pmc_fused.txt $D/pmc_f32_fused.txt
bash tools/pmc_bf16.sh 12 > /dev/null 2>&1; cp gpurun_out/pmc_bf16.txt $D/pmc_bf16.txt
echo "pmc done" >> $D/README.txt
# 3. kernel statistics of the driver's command (python3 directly behind --) and of 96 training steps of each executor
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_under_rocprof.json 2> $D/bench_under_rocprof.err; find /tmp/prof_final -name "*kernel_stats.csv" -exec cp {} $D/kernel_stats_driver_cmd.csv \; )
bash tools/prof_bf16.sh final_bf16 96 > $D/bf16_step_table.txt 2>&1; cp gpurun_out/prof_final_bf16/kernel_stats.csv $D/kernel_stats_bf16_steps.csv
bash tools/prof_bf16.sh final_f32 96 f32 > $D/f32_step_table.txt 2>&1; cp gpurun_out/prof_final_f32/kernel_stats.csv $D/kernel_stats_f32_steps.csv
echo "kernel stats done" >> $D/README.txt
# 4. the driver's command, unprofiled (after the counters: its roofline.traffic then quotes THIS build's bytes)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_cmd.json 2> $D/bench_driver_cmd.err
echo "bench rc=$?" >> $D/README.txt
python3 tools/bf16_train_speed.py --classes > $D/train_speed_classes.txt 2>&1
python3 tools/stage_split.py > $D/stage_split.txt 2>&1
tail -3 $D/README.txt
