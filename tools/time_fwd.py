"""Diagnostic: forward-only timings (save / no save, SIREN / FFN) to separate VALU from stash-store cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
torch.manual_seed(0)
enc = M.Positional_Encoder(enc_cfg, device=dev)
coords = (torch.rand(B, 3) * 2 - 1).to(dev)
for name, cls in (("SIREN", M.SIREN), ("FFN", M.FFN)):
    model = cls(net).to(dev)
    eng = model.fused_engine(256)
    for save in (False, True):
        for _ in range(3): eng.forward(coords, enc.B.contiguous(), save=save)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): eng.forward(coords, enc.B.contiguous(), save=save)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name} fwd save={save}: {ms*1e3:.1f} us  (MFMA-bound at 2.4 GHz: {5248*64/2.4e3:.1f} us)")
