"""Diagnostic (GPU): PSNR and mean batch loss of the benchmark fit (bench.CONFIG, synthetic k-space) every EVERY steps, fp32
and bf16 side by side; for bf16 also the gradient-scale state.   python tools/psnr_trajectory.py [steps] [every]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import bench
from inr_mi355x.synthetic import make_kspace
from inr_mi355x.train import INRTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
image, coords, shape = make_kspace(*bench.SHAPE, seed=1234, normalization="coil")
cfg = dict(bench.CONFIG); cfg["batch_size"] = 25000
out = {}
for prec in ("f32", "bf16"):
    c = dict(cfg)
    if prec == "bf16":
        c["precision"] = "bf16"
    tr = INRTrainer(c, image, coords, shape, dev, seed=0)
    spe = tr.steps_per_epoch
    rows, acc, worst = [], 0.0, 0.0
    for s in range(steps):
        l = tr.step(s // spe, s % spe)
        if (s + 1) % 50 == 0:  # (a host read every 50 steps: cheap enough, and catches spikes)
            v = float(l); acc += v; worst = max(worst, v)
        if (s + 1) % every == 0:
            st = tr.engine.grad_scale_state() if prec == "bf16" else None
            rows.append({"step": s + 1, "psnr_db": tr.evaluate(), "mean_loss": acc / (every / 50), "max_loss": worst,
                         "scale": None if st is None else [st[0], st[3]]})
            acc, worst = 0.0, 0.0
            print(prec, rows[-1], flush=True)
    out[prec] = rows
    del tr
print(json.dumps(out))
