"""Diagnostic: the row-split fused step (inr_mlp_rs_impl.h, default) against inr_mlp_kernel (INR_RS=0) on the same inputs:
loss, gradient and outputs, for ragged batches / depths / widths / encoder sizes; then HIP-event timings of both.

    python tools/debug_rs.py [--time]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mri-implicit-neural-representations_amd")]
import inr_mi355x as M
from inr_mi355x import _lib as L

dev = torch.device("cuda:0")


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def case(B, depth=5, width=256, E=256, masked=False, loss=L.LOSS_L2_HALF, out_f=2, kind="SIREN", last_tanh=False):
    net = dict(network_input_size=2 * E, network_output_size=out_f, network_depth=depth, network_width=width,
               last_tanh=last_tanh)
    enc_cfg = dict(embedding="gauss", scale=2, embedding_size=E, coordinates_size=3)
    torch.manual_seed(B + depth)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    model = (M.SIREN if kind == "SIREN" else M.FFN)(net).to(dev)
    g = torch.Generator().manual_seed(B)
    coords = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
    gt = (torch.randn(B, out_f, generator=g) * 0.2).to(dev)
    mask = (torch.rand(B, generator=g) < 0.6).to(torch.uint8).to(dev) if masked else None
    cnt = B if mask is None else int(mask.sum())
    eng = model.fused_engine(E)
    res = {}
    for rs in ("0", "1"):
        os.environ["INR_RS"] = rs
        eng.grads.zero_()
        l = eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(loss), count=cnt, mask=mask).clone()
        torch.cuda.synchronize()
        res[rs] = (l.cpu(), eng.grads.clone().cpu())
    os.environ["INR_RS"] = "1"
    l2 = eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(loss), count=cnt, mask=mask).clone()
    det = torch.equal(eng.grads.cpu(), res["1"][1]) and torch.equal(l2.cpu(), res["1"][0])
    # per-layer relative error
    offs, per = 0, []
    for p in model.parameters():
        n = p.numel()
        per.append(rel(res["1"][1][offs:offs + n], res["0"][1][offs:offs + n]))
        offs += n
    bad = not all(e < 2e-5 for e in per) or not torch.isfinite(res["1"][1]).all()
    print(f"B={B:6d} D={depth} W={width} E={E} mask={int(masked)} loss={loss} {kind} out={out_f}: loss {float(res['0'][0]):.7g} / "
          f"{float(res['1'][0]):.7g}  grad rel {rel(res['1'][1], res['0'][1]):.2e}  det={det}  "
          f"per-tensor max {max(per):.2e}{'   <<<<<< BAD' if bad or not det else ''}", flush=True)
    if bad:
        print("   per tensor:", " ".join(f"{e:.1e}" for e in per))
    return not bad and det


def timing(B, steps=50):
    net = dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=False)
    enc_cfg = dict(embedding="gauss", scale=2, embedding_size=256, coordinates_size=3)
    torch.manual_seed(0)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    model = M.SIREN(net).to(dev)
    coords = (torch.rand(B, 3) * 2 - 1).to(dev)
    gt = (torch.randn(B, 2) * 0.2).to(dev)
    eng = model.fused_engine(256)
    for rs in ("0", "1", "0", "1"):
        os.environ["INR_RS"] = rs
        for _ in range(10):
            eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} INR_RS={rs}: {e0.elapsed_time(e1) / steps:.4f} ms per gradient step", flush=True)


if __name__ == "__main__":
    ok = True
    if "--time" not in sys.argv:
        for B in (1, 16, 17, 127, 128, 129, 1000, 4133, 25000, 40000, 65536, 100000):
            ok &= case(B)
        ok &= case(4133, masked=True)
        ok &= case(25000, masked=True, loss=L.LOSS_HDR)
        for depth in (2, 3, 4, 8):
            ok &= case(3000, depth=depth)
        for width, E in ((160, 32), (200, 96), (256, 512), (129, 64)):
            ok &= case(2777, width=width, E=E)
        ok &= case(3000, kind="FFN")
        ok &= case(3000, out_f=3)
        ok &= case(3000, out_f=1, last_tanh=True)
        print("ALL OK" if ok else "FAILURES")
    timing(25000)
    timing(65536)
    sys.exit(0 if ok else 1)
