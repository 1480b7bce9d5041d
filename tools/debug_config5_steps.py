"""Diagnostic (GPU): per-step wall clock of the config-5 trainer (per-coil, radial mask, TV) -- where do slow rounds come from?
python tools/debug_config5_steps.py [bf16|f32] [steps]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch, yaml
import bench
from inr_mi355x.synthetic import make_kspace
from inr_mi355x.train import INRTrainer
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_siren_radial_tv_bf16.yaml")))
cfg.pop("precision")
if prec == "bf16":
    cfg["precision"] = "bf16"
image, coords, shape = make_kspace(2, bench.SHAPE[1], bench.SHAPE[2], seed=1234, normalization="coil")
tr = INRTrainer(cfg, image, coords, shape, dev, seed=0, mask_seed=7)
spe = tr.steps_per_epoch
for i in range(4):
    tr.step(0, i % spe)
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
ts = []
gc_before = gc.get_count()
for i in range(steps):
    t0 = time.perf_counter()
    tr.step(0, i % spe)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
st1 = torch.cuda.memory_stats()
import numpy as np
a = np.array(ts) * 1e3
tot = a.sum(1)
print(f"{prec}: {steps} steps; total per step: median {np.median(tot):.3f} ms, mean {tot.mean():.3f}, max {tot.max():.1f}")
print(f"   host (enqueue) part: median {np.median(a[:,0]):.3f} mean {a[:,0].mean():.3f} max {a[:,0].max():.1f};  wait part: median {np.median(a[:,1]):.3f} mean {a[:,1].mean():.3f} max {a[:,1].max():.1f}")
slow = np.nonzero(tot > 4 * np.median(tot))[0]
print(f"   {len(slow)} steps above 4x the median: indices {slow[:40].tolist()}")
print("   of those, (enqueue ms, wait ms):", [(round(a[i,0],1), round(a[i,1],1)) for i in slow[:12]])
for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "allocation.all.allocated", "segment.all.allocated"):
    print(f"   allocator {k}: {st0.get(k)} -> {st1.get(k)}")
print("   gc counts", gc_before, "->", gc.get_count(), " gc stats", gc.get_stats()[-1])
