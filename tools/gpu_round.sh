#!/bin/bash
# One GPU-box call of the round-2 routine: tests (all, no -x), phase stamps, bench on the shipped library and on
# the polynomial-sincos A/B build, secondary workloads.  Everything lands under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-r2}
rm -f gpurun_out/parity_errors.jsonl
echo "== tests" && timeout -k 10 600 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
echo "== bench (shipped lib)" && timeout -k 10 400 python bench.py --steps 50 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
if [ -f mri-implicit-neural-representations_amd/lib/libinr_mi355x_poly.so ]; then
  echo "== bench (poly sincos A/B)" && INR_LIB_PATH=$PWD/mri-implicit-neural-representations_amd/lib/libinr_mi355x_poly.so timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-bf16 --no-multiscale --psnr-steps 0 > gpurun_out/${TAG}_bench_poly.json 2> gpurun_out/${TAG}_bench_poly.err; echo "bench poly rc=$?"
fi
echo "== secondary workloads" && timeout -k 10 200 python tools/bench_models.py > gpurun_out/${TAG}_bench_models.json 2> gpurun_out/${TAG}_bench_models.err; echo "models rc=$?"
if [ -f mri-implicit-neural-representations_amd/lib/libinr_mi355x_dbg.so ]; then
  echo "== stamps" && timeout -k 10 120 python tools/stamps.py 25000 f32 > gpurun_out/${TAG}_stamps_f32.log 2>&1; echo "stamps f32 rc=$?"
  timeout -k 10 120 python tools/stamps.py 100000 mfn > gpurun_out/${TAG}_stamps_mfn.log 2>&1; echo "stamps mfn rc=$?"
  timeout -k 10 120 python tools/stamps.py 25000 bf16 > gpurun_out/${TAG}_stamps_bf16.log 2>&1; echo "stamps bf16 rc=$?"
  timeout -k 10 120 python tools/stamps.py 65536 bf16 > gpurun_out/${TAG}_stamps_bf16_65536.log 2>&1
  timeout -k 10 120 python tools/stamps.py 65536 f32 > gpurun_out/${TAG}_stamps_f32_65536.log 2>&1
  timeout -k 10 120 python tools/stamps.py 25000 wire > gpurun_out/${TAG}_stamps_wire.log 2>&1
fi
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/*_bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    if 'value' in d:
        print(f, 'value %.1f M/s, %.4f ms/step, fused frac %.3f (%.4f ms), path frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['gradient_path']['frac']), {k:(v if not isinstance(v,dict) else '...') for k,v in d.get('psnr_at_1k_steps',{}).items()})
        for k in ('batch_65536','bf16_path','multiscale_config4','cpu_baseline'):
            if k in d: print('   ',k, json.dumps(d[k])[:600])
    else:
        print(f, json.dumps(d)[:1500])
PY
