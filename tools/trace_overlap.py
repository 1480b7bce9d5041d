"""Diagnostic: WIRE 4x256 / B = 25 000 steps in a loop (run under rocprofv3 --kernel-trace), or, with a kernel-trace
CSV as argument, the timeline of the last step in it (start / end in microseconds relative to the step's first kernel)."""
import os, sys, csv
if len(sys.argv) > 1 and sys.argv[1].endswith(".csv"):
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # steps = groups ending with adam_pack_kernel
    adam = [i for i, n in enumerate(names) if "adam_pack_kernel" in n]
    lo, hi = adam[-2] + 1, adam[-1]
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi + 1]:
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        print("%9.1f %9.1f  q=%s wgs=%4d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                r.get("Queue_Id", "?"), g, r["Kernel_Name"][:70]))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import torch
import inr_mi355x as M
from inr_mi355x import _lib as L
dev = torch.device("cuda:0")
B = 25000
torch.manual_seed(0)
net = dict(network_input_size=3, network_output_size=2, network_depth=4, network_width=256, first_omega_0=30, hidden_omega_0=30, scale=15)
eng = M.WIRE(net).to(dev)._engine()
coords = (torch.rand(B, 3) * 2 - 1).to(dev); gt = (torch.randn(B, 2) * 0.2).to(dev)
for _ in range(30):
    eng.train_step(coords, None, gt, M.LossSpec(L.LOSS_HDR), hdr_A=0.3); eng.adam_step(1e-4)
torch.cuda.synchronize()
