#!/bin/bash
# wide-kernel iteration (WIRE / WIRE2D / SIREN 8x512): their tests, the secondary-workload bench, WIRE phase stamps
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-wire}
timeout -k 10 600 python -m pytest tests/test_gpu_wire.py tests/test_gpu_widths.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
timeout -k 10 300 python tools/bench_models.py > gpurun_out/${TAG}_bench_models.json 2> gpurun_out/${TAG}_bench_models.err; echo "models rc=$?"
python - "$TAG" <<'PY'
import json,sys
d=json.load(open('gpurun_out/'+sys.argv[1]+'_bench_models.json'))
for k,v in d.items(): print('%-42s %8.3f ms  frac %.3f'%(k, v['ms_per_step'], v['frac_f32_mfma']))
PY
timeout -k 10 120 python tools/stamps.py 25000 ${STAMP_MODE:-wire} 2>&1 | grep -v amdgpu.ids
