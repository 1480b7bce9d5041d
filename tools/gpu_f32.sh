#!/bin/bash
# fp32 path iteration: all GPU tests, the bench's fp32 objects, phase stamps
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-f32}
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-multiscale --no-bf16 --psnr-steps 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err
python - "$TAG" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/'+sys.argv[1]+'_bench.json').read().strip().splitlines()[-1])
print('f32 value %.2f M/s  step %.4f ms  fused %.4f ms frac %.3f  path frac %.3f'%(d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['gradient_path']['frac']))
print(json.dumps(d.get('batch_65536')))
PY
timeout -k 10 120 python tools/stamps.py 25000 f32 2>&1 | grep -v amdgpu.ids
