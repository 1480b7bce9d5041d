import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
from inr_mi355x.mfn import MultiscaleKFourier
from inr_mi355x.engine import ConsistencySpec
dev = torch.device("cuda:0")
B = 100000
torch.manual_seed(0)
net = dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512)
enc = M.Positional_Encoder(dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3), device=dev)
model = MultiscaleKFourier(net).to(dev).bind_encoder(enc); eng = model._engine()
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
spec = M.LossSpec(L.LOSS_LOGSPACE, 3e-3)
for _ in range(5):
    eng.train_step(coords, enc.B.contiguous(), gt, spec, dist=dist, scale=0.5); eng.adam_step(3e-4)
torch.cuda.synchronize()
net = dict(network_input_size=3, network_output_size=2, network_depth=4, network_width=256, first_omega_0=30, hidden_omega_0=30, scale=15)
model = M.WIRE(net).to(dev); eng = model._engine()
c2 = coords[:25000].contiguous(); g2 = gt[:25000].contiguous()
for _ in range(5):
    eng.train_step(c2, None, g2, M.LossSpec(L.LOSS_HDR), hdr_A=0.3); eng.adam_step(1e-4)
torch.cuda.synchronize()
