"""Diagnostic: run-to-run determinism of the bf16 fused step, per parameter tensor (prints the tensors whose gradient
bits differ between launches on the same inputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
dev = torch.device("cuda:0")
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
for B in [int(a) for a in sys.argv[1:]] or [32845, 25000, 65536, 4133]:
    torch.manual_seed(B)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    m = M.SIREN(net).to(dev)
    e = m.fused_engine(256, precision="bf16")
    g = torch.Generator().manual_seed(B)
    coords = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
    gt = (torch.randn(B, 2, generator=g) * 0.2).to(dev)
    mask = (torch.rand(B, generator=g) < 0.7).to(torch.uint8).to(dev)
    spec = M.LossSpec(L.LOSS_L2_HALF)
    encB = enc.B.contiguous()
    e.train_step(coords, encB, gt, spec, count=int(mask.sum()), mask=mask)
    ref = e.grads.clone(); lref = float(e._loss_word[0])
    bad = 0
    for it in range(30):
        e.train_step(coords, encB, gt, spec, count=int(mask.sum()), mask=mask)
        if not torch.equal(ref, e.grads) or float(e._loss_word[0]) != lref:
            bad += 1
            if bad <= 3:
                for (name, p_), (o, n, s_, c) in zip(m.named_parameters(), m._layout):
                    a, b = e.grads[o:o + n], ref[o:o + n]
                    if not torch.equal(a, b):
                        d = (a - b).abs()
                        print("  B=%d it=%d %s: %d of %d entries differ, max |d| %.3e (|g| max %.3e)" % (B, it, name, int((d > 0).sum()), n, float(d.max()), float(b.abs().max())))
    print("B=%d: %d of 30 repeats differ from the first launch" % (B, bad))
