"""Checker script, not a test module (GPU): the bf16 step against its rounding oracle, tensor by tensor -- lives under tests/
because it imports the oracle.  python tests/debug_bf16_oracle.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import numpy as np, torch
import inr_mi355x as M
from inr_mi355x import _lib as L
import oracle as O

NET = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
ENC = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


for B in [int(x) for x in sys.argv[1:]] or [127, 4133]:
    torch.manual_seed(B)
    enc = M.Positional_Encoder(ENC, device=dev)
    model = M.SIREN(NET)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    eng = model.fused_engine(256, precision="bf16")
    e32 = M.SIREN(NET).to(dev); e32.load_state_dict(model.state_dict()); e32 = e32.fused_engine(256)
    g = torch.Generator().manual_seed(B + 1)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    encB = enc.B.contiguous()
    out = eng.forward(coords.to(dev), encB).cpu()
    loss = float(eng.train_step(coords.to(dev), encB, gt.to(dev), M.LossSpec(L.LOSS_L2_HALF)))
    st = eng.grad_scale_state()
    y, ref, amax = O.bf16.siren_bf16_step(sd, coords, enc.B.cpu(), NET, lambda yy: (yy - gt) / (B * 2.0), st[2])
    l32 = float(e32.train_step(coords.to(dev), encB, gt.to(dev), M.LossSpec(L.LOSS_L2_HALF)))
    print(f"B={B} loss {loss:.6g} (fp32 engine {l32:.6g}) state {st} oracle amax {amax:.4g} out max err {float((out - y).abs().max()):.3g}")
    for (name, p_), (o, n, s_, c) in zip(model.named_parameters(), model._layout):
        got, want, g32 = eng.grads[o:o + n].cpu(), ref[name].reshape(-1), e32.grads[o:o + n].cpu()
        print(f"   {name:28s} vs oracle {rel(got, want):.3e}  |got|/|want| {float(got.norm() / (want.norm() + 1e-30)):.4f}   vs fp32 engine {rel(got, g32):.3e}")
    o, n = model._layout[0][0], model._layout[0][1]
    got, want = eng.grads[o:o + n].cpu().reshape(256, 512), ref["model.0.linear.weight"]
    print("   dW0 by column block of 64:", " ".join(f"{rel(got[:, c:c + 64], want[:, c:c + 64]):.1e}" for c in range(0, 512, 64)))
    print("   dW0 by row block of 32:   ", " ".join(f"{rel(got[c:c + 32], want[c:c + 32]):.1e}" for c in range(0, 256, 32)))
    import oracle.inr_oracle_bf16 as OB
    keep = OB._f16
    for nm, fn in (("bf16", OB._bf16), ("none", lambda x: x.float())):
        OB._f16 = fn
        _, ref2, _ = OB.siren_bf16_step(sd, coords, enc.B.cpu(), NET, lambda yy: (yy - gt) / (B * 2.0), st[2])
        print(f"   GEMM operands rounded to {nm}: dW0 device vs that {rel(got, ref2['model.0.linear.weight']):.2e}, dW1 "
              f"{rel(eng.grads[model._layout[2][0]:model._layout[2][0] + model._layout[2][1]].cpu(), ref2['model.1.linear.weight'].reshape(-1)):.2e}")
    OB._f16 = keep
    d = (got - want).abs()
    idx = torch.nonzero(d > 20 * d.mean())
    print("   outliers (>20x mean abs err):", idx.shape[0], "rows", sorted(set(idx[:, 0].tolist()))[:12], "cols", sorted(set(idx[:, 1].tolist()))[:24])
    l2 = float(eng.train_step(coords.to(dev), encB, gt.to(dev), M.LossSpec(L.LOSS_L2_HALF)))
    print("   second step state", eng.grad_scale_state(), "loss equal", l2 == loss)
