"""Diagnostic: which stash entries of the bf16 fused step differ between two launches on the same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
torch.manual_seed(B)
enc = M.Positional_Encoder(enc_cfg, device=dev)
m = M.SIREN(net).to(dev)
e = m.fused_engine(256, precision="bf16")
g = torch.Generator().manual_seed(B)
coords = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
gt = (torch.randn(B, 2, generator=g) * 0.2).to(dev)
spec = M.LossSpec(L.LOSS_L2_HALF)
encB = enc.B.contiguous()
e.train_step(coords, encB, gt, spec); torch.cuda.synchronize()
s0 = e._save.clone().view(torch.int32)
per_tile = e.save_floats_per_tile
T = 256 * 128 // 2  # dwords per stashed 2-byte tensor [256 rows][128 coordinates]
for it in range(5):
    e.train_step(coords, encB, gt, spec); torch.cuda.synchronize()
    s1 = e._save.view(torch.int32)
    nt = e.launch_dims(B)[0]
    d = (s0[: nt * per_tile] != s1[: nt * per_tile]).view(nt, per_tile)
    if not bool(d.any()):
        print("run %d: stash identical" % it); continue
    tiles = d.any(dim=1).nonzero().flatten().tolist()
    print("run %d: %d tiles differ (first %s)" % (it, len(tiles), tiles[:8]))
    t = tiles[0]
    idx = d[t].nonzero().flatten()
    tens = (idx // T).unique().tolist()
    print("   tile %d: tensors %s (dword index // %d); per tensor:" % (t, tens, T))
    for k in tens[:6]:
        sel = idx[(idx // T) == k] - k * T
        rows = (sel // 128).unique().tolist()   # dword = (row pair, coordinate): [128 row pairs][128 coordinates]
        cols = (sel % 128).unique().tolist()
        print("      tensor %d: %d dwords, row pairs %s..., coordinates %s..." % (k, len(sel), rows[:10], cols[:10]))
