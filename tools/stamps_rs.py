"""Diagnostic: per-phase cycle shares of the row-split fused step (needs tools/build_dbg_rs.sh).
Run on the GPU box:  python tools/stamps_rs.py [B] [depth]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mri-implicit-neural-representations_amd")
os.environ.setdefault("INR_LIB_PATH", os.path.join(PKG, "lib", "libinr_mi355x_dbg.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
torch.manual_seed(0)
enc = M.Positional_Encoder(enc_cfg, device=dev)
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
lib = L.load()
lib.inr_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_longlong]
net = dict(network_input_size=512, network_output_size=2, network_depth=D, network_width=256, last_tanh=True)
model = M.SIREN(net).to(dev)
eng = model.fused_engine(256)
ws = eng._ws(*eng.workspace(B)); ld = eng.loss_desc(M.LossSpec(L.LOSS_L2_HALF), B)
encB = enc.B.contiguous()


def fused_only():
    L.check(lib.inr_train_step(eng.plan, C.byref(ld), eng.params.data_ptr(), eng.packed.data_ptr(), coords.data_ptr(),
                               encB.data_ptr(), gt.data_ptr(), None, B, C.byref(ws), None, eng._loss_word.data_ptr(),
                               eng._stream()))


NB = 256
dbg = torch.zeros(NB * 4 * 64, dtype=torch.int64, device=dev)
for _ in range(200):
    fused_only()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    fused_only()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
lib.inr_debug_set_stamp_buffer(dbg.data_ptr(), dbg.numel())
import time
t_warm = time.perf_counter()  # >= 2.5 s of back-to-back launches before the one whose stamps stay (DVFS settles in seconds)
while time.perf_counter() - t_warm < 2.5:
    for _ in range(200):
        fused_only()
    torch.cuda.synchronize()
lib.inr_debug_set_stamp_buffer(None, 0)
d = dbg.cpu().view(NB, 4, 64).double()
live = d[:, 0, 40] > 0
d = d[live]
order = [(0, "start"), (1, "layer 0: features + GEMM")]
for l in range(1, D - 1):
    order += [(20 + l - 1, f"epilogue L{l-1} + syncs"), (1 + l, f"GEMM fwd L{l}")]
order += [(10, "sync"), (20 + D - 2, "last hidden epilogue + last layer + loss + adjoint")]
for l in range(D - 2, 0, -1):
    order += [(12 + l, f"GEMM bwd L{l}")]
    if l > 1:
        order += [(30 + l, f"epilogue dZ_{l-1} + syncs")]
order += [(40, "dZ_0 -> stash")]
tot = d[:, :, 40] - d[:, :, 0]
rt = d[:, :, 63] - d[:, :, 62]
ghz = (tot / rt * 0.1)[rt > 0]
rt0, rt1 = d[:, :, 62][d[:, :, 62] > 0], d[:, :, 63][d[:, :, 63] > 0]
print(f"100 MHz counter, all waves of the launch: first stamp of the earliest wave -> last stamp of the latest "
      f"{(rt1.max() - rt0.min()) / 100:.1f} us; first stamps spread over {(rt0.max() - rt0.min()) / 100:.1f} us, last stamps over "
      f"{(rt1.max() - rt1.min()) / 100:.1f} us")
print(f"in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz, first to last stamp of a wave, after >= 2.5 s of launches): "
      f"median {ghz.median():.3f} GHz, min {ghz.min():.3f}, max {ghz.max():.3f}")
print(f"B={B} D={D}: fused kernel alone {us:.1f} us per launch; workgroups with a tile {int(live.sum())}; last tile of each: "
      f"cycles/wave mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}  -> >= {tot.max() / us / 1e3:.2f} GHz")
prev = 0
for i, name in order[1:]:
    seg = d[:, :, i] - d[:, :, prev]
    print(f"  {name:>52s}: mean {seg.mean():9.0f}  per-wave {[round(float(seg[:, w].mean())) for w in range(4)]}  max {seg.max():.0f}")
    prev = i

def seg(a, b, name):
    x = d[:, :, b] - d[:, :, a]
    print(f"  {name:>52s}: mean {x.mean():9.0f}  per-wave {[round(float(x[:, w].mean())) for w in range(4)]}  max {x.max():.0f}")
print("inside the last-layer section:")
seg(10, 50, "last hidden epilogue + partial outputs")
seg(50, 51, "quarter sums (shuffles) + partials -> LDS")
seg(51, 52, "barrier")
seg(52, 53, "loss lanes")
seg(53, 54, "barrier")
seg(54, 55, "dW_last, dZ_{D-2} -> image")
print("layer 0, chunk 1:")
seg(56, 57, "features of chunk 2")
seg(57, 58, "GEMM of chunk 1 (16 k-steps)")
seg(58, 59, "barrier")
print("hidden layer 1 epilogue:")
seg(2, 61, "barrier")
seg(61, 60, "activation + image rows")
seg(60, 21, "barrier")
