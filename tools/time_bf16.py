"""Diagnostic: the bf16 (or f32) gradient path + Adam at a given batch in a loop (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
enc = M.Positional_Encoder(dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3), device=dev)
eng = M.SIREN(net).to(dev).fused_engine(256, precision=prec)
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
for _ in range(60):
    eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF)); eng.adam_step(3e-5)
torch.cuda.synchronize()
