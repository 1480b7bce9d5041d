#!/bin/bash
# Kernel durations of the bf16 path from rocprofv3 (run on the GPU box from the repo root): tools/prof_bf16.sh TAG [B ...]
set -o pipefail
R=${1:-bf16}; shift
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_kt -o ${R} -- python3 $ROOT/tools/bench_bf16.py "$@" > $ROOT/gpurun_out/${R}_kt.log 2>&1) || { echo "kernel-trace pass failed"; tail -5 gpurun_out/${R}_kt.log; exit 1; }
kt=$(find gpurun_out/${R}_kt -name '*kernel_trace.csv' | head -1)
python3 tools/kernel_stats_by_grid.py "$kt" > gpurun_out/${R}_kernel_stats_by_grid.csv
rm -rf gpurun_out/${R}_kt
cat gpurun_out/${R}_kernel_stats_by_grid.csv | cut -c1-200 | head -20
tail -2 gpurun_out/${R}_kt.log
