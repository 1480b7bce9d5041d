#!/bin/bash
# PMC passes over the bf16 path (run on the GPU box from the repo root): each pass its own process, kernel-trace only.
# usage: tools/pmc_bf16.sh TAG [B]   ->  gpurun_out/TAG_pmc_by_grid.csv
set -o pipefail
R=${1:-bf16pmc}; B=${2:-65536}
export TMPDIR=/tmp
ROOT=$(pwd)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $set -d $ROOT/gpurun_out/${R}_pmc_$i -o p -- python3 $ROOT/tools/time_bf16.py $B bf16 > $ROOT/gpurun_out/${R}_pmc_$i.log 2>&1) || { echo "pass $i ($set) failed"; tail -3 gpurun_out/${R}_pmc_$i.log; }
done
python3 tools/pmc_by_grid.py $(find gpurun_out/${R}_pmc_* -name '*counter_collection.csv') > gpurun_out/${R}_pmc_by_grid.csv
rm -rf gpurun_out/${R}_pmc_[0-9]
grep -E "siren_bf16|dw_gemm_bf16" gpurun_out/${R}_pmc_by_grid.csv | cut -c1-160
