#!/bin/bash
# bf16 path iteration: its tests + the bench's bf16 object only
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-bf16}
timeout -k 10 400 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -m gpu -q -x -k "bf16" > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/${TAG}_tests.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-multiscale --psnr-steps 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err
python - "$TAG" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/'+sys.argv[1]+'_bench.json').read().strip().splitlines()[-1])
    print('f32 value %.1f M/s frac %.3f'%(d['value']/1e6, d['roofline']['frac']))
    print(json.dumps(d.get('bf16_path'),indent=1))
except Exception as e: print('no bench', e)
PY
