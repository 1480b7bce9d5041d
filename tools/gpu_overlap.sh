#!/bin/bash
# split steps (GEMM part A beside the fused kernel's partial round): tests, then the secondary workloads both ways
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-ovl}
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "split_step or persistent_blocks or wire_full_size or full_baseline or trajectory_golden" > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/${TAG}_tests.log
for O in 0 1; do
  INR_OVERLAP=$O timeout -k 10 300 python tools/bench_models.py > gpurun_out/${TAG}_models_o$O.json 2> gpurun_out/${TAG}_models_o$O.err; echo "models overlap=$O rc=$?"
  python - "$TAG" $O <<'PY'
import json,sys
d=json.load(open('gpurun_out/%s_models_o%s.json'%(sys.argv[1],sys.argv[2])))
print('overlap=%s'%sys.argv[2], {k:(round(v['ms_per_step'],4), round(v['frac_f32_mfma'],4)) for k,v in d.items()})
PY
done
