#!/bin/bash
# two training processes on one GPU at the same time (tools/concurrency_probe.py), then one alone
cd "$(dirname "$0")/.."
timeout -k 10 300 python tools/concurrency_probe.py A ${1:-6} > gpurun_out/conc_A.txt 2>&1 &
PA=$!
timeout -k 10 300 python tools/concurrency_probe.py B ${1:-6} > gpurun_out/conc_B.txt 2>&1 &
PB=$!
wait $PA; wait $PB
timeout -k 10 300 python tools/concurrency_probe.py alone 3 > gpurun_out/conc_alone.txt 2>&1
grep -h "round\|differing\|Error" gpurun_out/conc_A.txt gpurun_out/conc_B.txt gpurun_out/conc_alone.txt
