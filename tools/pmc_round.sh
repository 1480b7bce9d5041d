#!/bin/bash
# Round measurements (run on the GPU box from the repo root): rocprofv3 kernel-trace stats of the bench command, then
# one --pmc pass per counter set (each its own process, kernel-trace only), then the per-(kernel, grid) summaries.
set -o pipefail
R=${1:-r02}
export TMPDIR=/tmp
ROOT=$(pwd)
B="bench.py --no-cpu-baseline --psnr-steps 0 --no-multiscale"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_kt -o ${R} -- python3 $ROOT/$B > $ROOT/gpurun_out/${R}_kt.log 2>&1) || { echo "kernel-trace pass failed"; tail -5 gpurun_out/${R}_kt.log; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_MFMA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $set -d $ROOT/gpurun_out/${R}_pmc_$i -o ${R} -- python3 $ROOT/$B --steps 20 --warmup 3 > $ROOT/gpurun_out/${R}_pmc_$i.log 2>&1) || { echo "pmc pass $i ($set) failed"; tail -3 gpurun_out/${R}_pmc_$i.log; }
done
kt=$(find gpurun_out/${R}_kt -name '*kernel_trace.csv' | head -1)
st=$(find gpurun_out/${R}_kt -name '*kernel_stats.csv' | head -1)
python3 tools/kernel_stats_by_grid.py "$kt" > gpurun_out/${R}_kernel_stats_by_grid.csv
cp "$st" gpurun_out/${R}_kernel_stats_bench.csv
python3 tools/pmc_by_grid.py $(find gpurun_out/${R}_pmc_* -name '*counter_collection.csv') > gpurun_out/${R}_pmc_by_grid.csv
rm -rf gpurun_out/${R}_kt gpurun_out/${R}_pmc_[0-9]
head -12 gpurun_out/${R}_kernel_stats_by_grid.csv; grep "FETCH_SIZE\|WRITE_SIZE" gpurun_out/${R}_pmc_by_grid.csv | head -30
