#!/bin/bash
# PMC passes for the bf16 fused kernel (run on the GPU box from the repo root): each pass its own process
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MFMA" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/pmc_bf16_$i -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --psnr-steps 0 > gpurun_out/pmc_bf16_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_bf16_*/p_counter_collection.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bf16_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 196 * 256:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        print(f, k, sum(v) / len(v), len(v))
PY
