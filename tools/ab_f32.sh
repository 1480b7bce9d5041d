#!/bin/bash
# A/B of experiment builds on the fp32 bench objects: bash tools/ab_f32.sh NAME [NAME...] (lib/libinr_exp_NAME.so, or "base")
ROOT=$(pwd)
for v in "$@"; do
  if [ "$v" = base ]; then unset INR_LIB_PATH; else export INR_LIB_PATH=$ROOT/mri-implicit-neural-representations_amd/lib/libinr_exp_$v.so; fi
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-multiscale --no-bf16 --psnr-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v: %.2f M/s step %.4f ms fused %.4f ms frac %.3f | 65536: fused %.4f frac %.3f'%(d['value']/1e6,d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'],d['batch_65536']['kernel_ms'],d['batch_65536']['frac']))"
done
