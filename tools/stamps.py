"""Diagnostic: per-phase cycle shares of the fused kernel (needs `make -C csrc dbg`).
Run on the GPU box:  INR_LIB_PATH=.../lib/libinr_mi355x_dbg.so python tools/stamps.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mri-implicit-neural-representations_amd")
os.environ.setdefault("INR_LIB_PATH", os.path.join(PKG, "lib", "libinr_mi355x_dbg.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
torch.manual_seed(0)
enc = M.Positional_Encoder(enc_cfg, device=dev)
model = M.SIREN(net).to(dev)
eng = model.fused_engine(256, precision=PREC)
lib = L.load()
lib.inr_debug_set_stamp_buffer.argtypes = [C.c_void_p]
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
nt, nb = eng.launch_dims(B)
NWV = eng.tile_rows // 32
dbg = torch.zeros(nb * NWV * 64, dtype=torch.int64, device=dev)
lib.inr_debug_set_stamp_buffer(dbg.data_ptr())
for _ in range(3):
    eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
torch.cuda.synchronize()
d = dbg.cpu().view(nb, NWV, 64).double()
names = {0: "start", 1: "fwd L0", 2: "fwd L1", 3: "fwd L2", 4: "fwd L3", 10: "fwd last+loss", 11: "sync", 12: "dW last",
         13: "dX last+store", 26: "dX L3", 29: "dZ3 -> stash", 22: "dX L2", 25: "dZ2 -> stash", 18: "dX L1",
         21: "dZ1 -> stash", 41: "dZ0 -> stash", 42: "sync"}
# (the 256-row builds leave dW of the hidden-width layers to inr_dw_gemm.hip: no dW phases in the kernel)
order = [0, 1, 2, 3, 4, 10, 11, 12, 13, 26, 29, 22, 25, 18, 21, 41, 42]
if PREC != "f32":
    names.update({27: "sync", 28: "dW L3", 29: "sync+store", 23: "sync", 24: "dW L2", 25: "sync+store", 19: "sync",
                  20: "dW L1", 21: "sync+store", 40: "dz0+sync", 41: "dW L0"})
    order = [0, 1, 2, 3, 4, 10, 11, 12, 13, 26, 27, 28, 29, 22, 23, 24, 25, 18, 19, 20, 21, 40, 41, 42]
tot = (d[:, :, 42] - d[:, :, 0])
print(f"B={B} blocks={nb} total cycles/wave: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
prev = order[0]
for i in order[1:]:
    seg = d[:, :, i] - d[:, :, prev]
    print(f"  {names[i]:>14s}: mean {seg.mean():9.0f}  per-wave means {[round(float(seg[:, w].mean())) for w in range(NWV)]}")
    prev = i
