"""Diagnostic: where a workgroup of the bf16 dW GEMM spends its cycles (needs an experiment build with -DGB_STAMPS:
tools/build_exp.sh stamps "-DGB_STAMPS"; run with INR_LIB_PATH=.../libinr_exp_stamps.so).  Per wave the kernel sums the
cycles of: [0] loop overhead / wait for set, [1] staging, [2] LDS reads + MFMA issue, [3] refill issue, [4] barrier,
[5] epilogue; prints them per unit kind (first layer / hidden / last) and wave half, averaged over workgroups."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import numpy as np, torch
import inr_mi355x as M
from inr_mi355x import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True)
enc = M.Positional_Encoder(dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3), device=dev)
eng = M.SIREN(net).to(dev).fused_engine(256, precision="bf16")
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
for _ in range(20):
    eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
torch.cuda.synchronize()
lib = L.load()
buf = np.zeros(512 * 8 * 8, dtype=np.int64)
rc = lib.inr_debug_gemm_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
s = buf.reshape(512, 8, 8)
n_units = net["network_depth"] + 1
nb = int((s[:, 0, 7] > 0).sum())
print(f"B {B}: {nb} workgroups with stamps, stages per workgroup {sorted(set(s[:nb, 0, 7].tolist()))}")
names = ["wait/loop", "staging", "reads+mfma", "refill", "barrier", "epilogue"]
for u in range(n_units):
    blk = s[u:nb:n_units]
    for half, sel in (("waves0-3", slice(0, 4)), ("waves4-7", slice(4, 8))):
        m = blk[:, sel, :6].mean(axis=(0, 1))
        tot = m.sum()
        print(f"unit {u} {half}: total {tot:9.0f} cyc  " + "  ".join(f"{n} {v:8.0f}" for n, v in zip(names, m)))
