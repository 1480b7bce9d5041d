#!/bin/bash
# Diagnostic: time the bf16 gradient path with experiment builds of the library (lib/libinr_exp_*.so), kernel by kernel.
# usage: bash tools/exp_libs.sh B NAME [NAME...]   (NAME = suffix of lib/libinr_exp_NAME.so, or "base")
set -o pipefail
B=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
for v in "$@"; do
  if [ "$v" = base ]; then export INR_LIB_PATH=$ROOT/mri-implicit-neural-representations_amd/lib/libinr_mi355x.so
  else export INR_LIB_PATH=$ROOT/mri-implicit-neural-representations_amd/lib/libinr_exp_$v.so; fi
  rm -rf /tmp/exp_$v
  # (the run's output is kept under gpurun_out/ whatever happens: a faulting experiment leaves its log behind)
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/exp_$v -o x -- python3 $ROOT/tools/time_bf16.py $B ${PREC:-bf16} > $ROOT/gpurun_out/exp_$v.log 2>&1) || { echo "$v failed"; tail -5 $ROOT/gpurun_out/exp_$v.log; exit 1; }
  f=$(find /tmp/exp_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0][:60]
    print(f"  {n:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  cp "$f" gpurun_out/exp_${v}_kernel_stats.csv
done
