import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-implicit-neural-representations_amd"))
import inr_mi355x as M
from inr_mi355x import _lib as L
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "model_meta.json")))
dev = torch.device("cuda:0")
def _t(a): return torch.from_numpy(np.asarray(a))
def rel(a, b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
for name in sys.argv[1:] or ["SIREN"]:
    meta = META[name]; arrs = dict(np.load(os.path.join(GOLD, f"model_{name}.npz")))
    cls = {"SIREN": M.SIREN, "FFN": M.FFN}[meta["model"]]
    torch.manual_seed(meta["seed"])
    enc = M.Positional_Encoder(meta["encoder"], device=dev) if meta["encoder"] else None
    model = cls(meta["net"]).to(dev)
    x, gt, coords = _t(arrs["x"]).to(dev), _t(arrs["gt"]).to(dev), _t(arrs["coords"]).to(dev)
    out = model(x); loss = 0.5*torch.nn.functional.mse_loss(out, gt); loss.backward()
    print(name, "tier1 out rel", rel(out.detach().cpu(), _t(arrs["out"])), "loss", float(loss), float(arrs["loss"]))
    for k, p in model.named_parameters():
        ref = _t(arrs["grad/"+k]); g = p.grad.cpu()
        print("  tier1", k, "rel", rel(g, ref), "|g|", float(g.abs().max()), "|ref|", float(ref.abs().max()))
    if enc is not None:
        eng = model.fused_engine(meta["encoder"]["embedding_size"])
        l = eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF))
        print(name, "fused loss", float(l))
        flat = eng.grads.cpu()
        for (off, n, shp, _c), (k, _) in zip(model._layout, model.named_parameters()):
            ref = _t(arrs["grad/"+k]); g = flat[off:off+n].view(shp)
            print("  fused", k, "rel", rel(g, ref), "|g|", float(g.abs().max()))
