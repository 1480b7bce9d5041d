#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd); mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ovl_kt -o ovl -- python3 $ROOT/tools/trace_overlap.py > $ROOT/gpurun_out/ovl_kt.log 2>&1) || { tail -5 gpurun_out/ovl_kt.log; exit 1; }
kt=$(find gpurun_out/ovl_kt -name '*kernel_trace.csv' | head -1)
python3 tools/trace_overlap.py "$kt" | tee gpurun_out/ovl_timeline.txt
rm -rf gpurun_out/ovl_kt
