"""Diagnostic (GPU): bf16 path at large batches and over a training run.  python tools/debug_bf16_train.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import numpy as np, torch
import inr_mi355x as M
from inr_mi355x import _lib as L
import bench
from inr_mi355x.synthetic import make_kspace
from inr_mi355x.train import INRTrainer

dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))

NET, ENC = bench.CONFIG["net"], bench.CONFIG["encoder"]
torch.manual_seed(0)
enc = M.Positional_Encoder(ENC, device=dev)
m32 = M.SIREN(NET).to(dev); m16 = M.SIREN(NET).to(dev); m16.load_state_dict(m32.state_dict())
e32, e16 = m32.fused_engine(256), m16.fused_engine(256, precision="bf16")
encB = enc.B.contiguous()
for B in (65536, 200000, 300001):
    g = torch.Generator().manual_seed(B)
    coords = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev); gt = (torch.randn(B, 2, generator=g) * 0.2).to(dev)
    o32, o16 = e32.forward(coords, encB), e16.forward(coords, encB)
    print(f"B={B}: forward max diff {float((o32 - o16).abs().max()):.3g} nan {bool(torch.isnan(o16).any())}", flush=True)
    l32 = float(e32.train_step(coords, encB, gt, M.LossSpec(L.LOSS_L2_HALF)))
    l16 = float(e16.train_step(coords, encB, gt, M.LossSpec(L.LOSS_L2_HALF)))
    print(f"   loss {l16:.6g} vs {l32:.6g}; grads rel {rel(e16.grads, e32.grads):.3g} nan {bool(torch.isnan(e16.grads).any())} state {e16.grad_scale_state()[:4]}", flush=True)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
image, coords, shape = make_kspace(*bench.SHAPE, seed=1234, normalization="coil")
tr = INRTrainer(dict(bench.CONFIG, precision="bf16"), image.to(dev), coords.to(dev), shape, dev, seed=0)
spe = tr.steps_per_epoch
for i in range(steps):
    loss = tr.step(i // spe, i % spe)
    if i % 20 == 0 or i in (140, 141, 142, 143):
        st = tr.engine.grad_scale_state()
        print(f"step {i} (batch {i % spe}) loss {float(loss):.6g} S {st[0]:.4g} mult {st[2]:.4g} params nan {bool(torch.isnan(tr.engine.params).any())} grads nan {bool(torch.isnan(tr.engine.grads).any())} |g| {float(tr.engine.grads.norm()):.3g}", flush=True)
        if torch.isnan(tr.engine.params).any():
            break
print("psnr", tr.evaluate())
