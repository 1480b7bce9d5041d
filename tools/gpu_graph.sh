#!/bin/bash
# graph-replayed steps vs eager launches: the new test, then the bench both ways (fp32 + bf16 step objects)
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-graph}
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "device_resident or trajectory_golden or library_is" > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/${TAG}_tests.log
for G in 0 1; do
timeout -k 10 300 python bench.py --steps 300 --warmup 20 --graph $G --no-cpu-baseline --no-multiscale --psnr-steps 0 > gpurun_out/${TAG}_bench_g$G.json 2> gpurun_out/${TAG}_bench_g$G.err; echo "bench g=$G rc=$?"; tail -3 gpurun_out/${TAG}_bench_g$G.err
python - "$TAG" $G <<'PY'
import json,sys
d=json.loads(open('gpurun_out/%s_bench_g%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print('graph=%s f32 value %.2f M/s  step %.4f ms  launch %s'%(sys.argv[2], d['value']/1e6, d['ms_per_step'], d['config'].get('launch')))
b=d.get('bf16_path') or {}
print('   bf16', {k:b[k] for k in b if k in ('value','ms_per_step','coord_samples_per_s')})
PY
done
