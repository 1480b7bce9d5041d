#!/bin/bash
# Round measurements of the fp32 path (run on the GPU box from the repo root): kernel-trace stats of the bench
# command, then one rocprofv3 --pmc pass per counter set (each its own process, kernel-trace only).
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_kt -o ${R} -- python3 bench.py --no-cpu-baseline --psnr-steps 0 > gpurun_out/${R}_kt.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/${R}_pmc_$name -o ${R} -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-bf16 --psnr-steps 0 > gpurun_out/${R}_pmc_$name.log 2>&1
done
ls gpurun_out/${R}_kt gpurun_out/${R}_pmc_*
