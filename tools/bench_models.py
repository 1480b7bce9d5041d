"""Secondary workloads (BASELINE configs 3 and 4): fused-step timings + algorithmic TFLOP/s."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
from inr_mi355x.mfn import MultiscaleKFourier, FourierNet
from inr_mi355x.engine import ConsistencySpec
dev = torch.device("cuda:0")
res = {}

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# config 3: WIRE depth 4 / width 256 (181 complex), HDR, B = 25000
B = 25000
torch.manual_seed(0)
net = dict(network_input_size=3, network_output_size=2, network_depth=4, network_width=256, first_omega_0=30, hidden_omega_0=30, scale=15)
model = M.WIRE(net).to(dev); eng = model._engine()
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
spec = M.LossSpec(L.LOSS_HDR)
def step():
    eng.train_step(coords, None, gt, spec, hdr_A=0.3); eng.adam_step(1e-4)
ms = timeit(step)
res["WIRE_4x256_HDR_B25000"] = {"ms_per_step": ms, "samples_per_s": B / ms * 1e3, "TFLOPs": 3155916 * B / ms / 1e9, "frac_f32_mfma": 3155916 * B / ms / 1e9 / 157.3}

# config 4: MultiscaleKFourier 8x512, LSL + consistency, B = 100000
B = 100000
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
net = dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512)
torch.manual_seed(0)
enc = M.Positional_Encoder(enc_cfg, device=dev)
model = MultiscaleKFourier(net).to(dev).bind_encoder(enc); eng = model._engine()
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
pairs = [(0.0, 0.2), (0.0, 0.45), (0.0, 0.8), (0.0, 5.0)]
cons = ConsistencySpec(0.1, pairs, [1e-5, 1e-5, 1e-5, 0.0], 2)
spec = M.LossSpec(L.LOSS_LOGSPACE, eps=3e-3)
def step2():
    eng.train_step(coords, enc.B.contiguous(), gt, spec, dist=dist, scale=0.5, cons=cons); eng.adam_step(3e-4)
ms = timeit(step2, n=5, warm=2)
res["MultiscaleKFourier_8x512_LSL_B100000"] = {"ms_per_step": ms, "samples_per_s": B / ms * 1e3, "TFLOPs": 19423232 * B / ms / 1e9, "frac_f32_mfma": 19423232 * B / ms / 1e9 / 157.3}

# the reference's shipped SIREN config (config/remote/config_siren_kspace.yaml): depth 8 / width 512, batch 100 000
B = 100000
torch.manual_seed(0)
net = dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512, last_tanh=True)
enc = M.Positional_Encoder(dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3), device=dev)
model = M.SIREN(net).to(dev); eng = model.fused_engine(256)
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
spec = M.LossSpec(L.LOSS_L2_HALF)
def step():
    eng.train_step(coords, enc.B.contiguous(), gt, spec); eng.adam_step(3e-5)
ms = timeit(step, n=10)
mac = 512 * 512 + 6 * 512 * 512 + 512 * 2
flop = 2 * (3 * mac - 512 * 512)
res["SIREN_8x512_L2_B100000"] = {"ms_per_step": ms, "samples_per_s": B / ms * 1e3, "TFLOPs": flop * B / ms / 1e9,
                                 "frac_f32_mfma": flop * B / ms / 1e9 / 157.3}
# the reference's shipped WIRE2D config (config/remote/config_wire2d_kspace.yaml): depth 3 / width 256 (not reduced)
B = 25000
torch.manual_seed(0)
net = dict(network_input_size=3, network_output_size=2, network_depth=3, network_width=256, first_omega_0=30, hidden_omega_0=30, scale=15)
model = M.WIRE2D(net).to(dev); eng = model._engine()
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
spec = M.LossSpec(L.LOSS_L2_HALF)
def step():
    eng.train_step(coords, None, gt, spec); eng.adam_step(1e-4)
ms = timeit(step, n=10)
mac = 4 * (2 * (3 * 256) // 4 * 4 // 4) + 0  # first layer: real 3 x 256, two Linears (negligible)
cmac = 3 * 2 * 256 * 256 + 256 * 2             # complex MACs: 3 hidden layers x 2 Linears + last
flop = 8 * cmac * 3                            # 8 FLOP per complex MAC, fwd + dW + dX
res["WIRE2D_3x256_L2_B25000"] = {"ms_per_step": ms, "samples_per_s": B / ms * 1e3, "TFLOPs": flop * B / ms / 1e9,
                                 "frac_f32_mfma": flop * B / ms / 1e9 / 157.3}
# BASELINE config 5: radial acc-4 mask, per-coil batches (235 520 coordinates), total-variation term -- bench.py's object
# (its timing loop repeats rounds until they agree: see there)
import bench, gc
del eng, model, coords, gt
gc.collect(); torch.cuda.empty_cache()
res["config5_percoil_tv"] = bench.config5_percoil(dev)
print(json.dumps(res, indent=1))
