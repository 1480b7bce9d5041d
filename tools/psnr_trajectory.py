"""Diagnostic (GPU): the smooth PSNR criterion of bench.py (mean over the last full epoch's end-of-coil reads, bench.psnr_read_steps)
at several step counts of ONE fit of the benchmark workload (bench.CONFIG, synthetic k-space), fp32 and bf16 side by side;
for bf16 also the gradient-scale counters (clipped / flushed steps).  The fp32 engine stands in for the reference here (it
walks the CPU oracle's trajectory to 1e-5; bench.py checks that at 1 000 steps with the oracle itself).

    python tools/psnr_trajectory.py [STEPS ...]     (default: 1000 5000)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import bench
from inr_mi355x.synthetic import make_kspace
from inr_mi355x.train import INRTrainer

totals = sorted(int(a) for a in sys.argv[1:]) or [1000, 5000]
dev = torch.device("cuda:0")
image, coords, shape = make_kspace(*bench.SHAPE, seed=1234, normalization="coil")
cfg = dict(bench.CONFIG); cfg["batch_size"] = 25000
reads, coil = set(), {}
runs = {}
for prec in ("f32", "bf16"):
    c = dict(cfg)
    if prec == "bf16":
        c["precision"] = "bf16"
    tr = INRTrainer(c, image, coords, shape, dev, seed=0)
    for t in totals:
        r, coil[t] = bench.psnr_read_steps(t, tr.steps_per_epoch, tr.bs, bench.SHAPE[1] * bench.SHAPE[2], bench.SHAPE[0])
        reads |= set(r)
    runs[prec] = bench.fit_with_reads(tr, 0, totals[-1], reads)
    if prec == "bf16":
        st = tr.engine.grad_scale_state()
        print(f"bf16 gradient-scale counters after {totals[-1]} steps: clipped steps {st[8]:.0f}, flushed steps {st[9]:.0f}")
    del tr
out = {}
for t in totals:
    a = bench.psnr_summary(runs["f32"], coil[t])
    b = bench.psnr_summary(runs["bf16"], coil[t], runs["f32"])
    out[str(t)] = {"f32": a, "bf16": b}
    v = b.get("vs_reference", {})
    print(f"{t:6d} steps: last-epoch mean PSNR fp32 {a.get('mean_db', float('nan')):.4f} dB (std {a.get('std_db', 0):.3f}), "
          f"bf16 {b.get('mean_db', float('nan')):.4f} dB (std {b.get('std_db', 0):.3f}); bf16 - fp32: mean {v.get('delta_of_means_db', float('nan')):+.4f} dB, "
          f"std of paired deltas {v.get('std_of_deltas_db', 0):.4f}, max |delta| {v.get('max_abs_delta_db', 0):.4f}; "
          f"single reads at step {t}: fp32 {runs['f32'][t]:.4f}, bf16 {runs['bf16'][t]:.4f} ({runs['bf16'][t] - runs['f32'][t]:+.4f})")
print(json.dumps(out))
