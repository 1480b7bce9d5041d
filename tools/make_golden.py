#!/usr/bin/env python3
"""Generate golden vectors under tests/golden/ by IMPORTING the reference's own model and
loss classes from /root/reference/src (build container only; the reference never travels).

What is written is data only: seeds, config dicts, input tensors and the outputs the reference
produced for them.  Run:  python tools/make_golden.py
"""
import hashlib
import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

sys.dont_write_bytecode = True  # the reference tree is read-only: importing it must not leave __pycache__ behind
sys.path.insert(0, REF)
# metrics/losses.py imports fastmri at module scope but only RadialL2Loss (unused) calls it.
_fm = types.ModuleType("fastmri")
_fm.complex_abs = lambda x: (x ** 2).sum(-1).sqrt()
sys.modules["fastmri"] = _fm

import contextlib, io  # noqa: E402

from models.networks import SIREN, WIRE, FFN, Positional_Encoder  # noqa: E402
from models.mfn import (FourierNet, GaborNet, KGaborNet, MultiscaleKFourier,  # noqa: E402
                        MultiscaleBoundedFourier)
from models.wire2d import WIRE2D  # noqa: E402
from models.regularization import Regularization_L1, Regularization_L2  # noqa: E402
from metrics.losses import (HDRLoss_FF, TanhL2Loss, LogSpaceLoss, ConsistencyLoss, CenterLoss,  # noqa: E402
                            tv_loss, MSLELoss)

torch.set_num_threads(1)  # deterministic reduction order


def npy(t):
    t = t.detach()
    if t.is_complex():
        t = torch.view_as_real(t)
    return t.cpu().numpy().copy()  # copy: state_dict tensors alias live parameters


def sha(t):
    return hashlib.sha256(npy(t).tobytes()).hexdigest()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


BOUNDS4 = [(0.0, 0.2), (0.0, 0.45), (0.0, 0.8), (0.0, 5.0)]
BOUNDS8 = [b for b in BOUNDS4 for _ in range(2)]

CTORS = {
    "SIREN": SIREN, "FFN": FFN, "WIRE": WIRE, "WIRE2D": WIRE2D, "Fourier": FourierNet,
    "Gabor": GaborNet, "KGabor": KGaborNet, "MultiscaleKFourier": MultiscaleKFourier,
    "BoundedFourier": lambda net: MultiscaleBoundedFourier(net, boundaries=BOUNDS8),
}

TINY = {
    "SIREN": dict(network_input_size=16, network_output_size=2, network_depth=4, network_width=32),
    "SIREN_tanh": dict(network_input_size=16, network_output_size=2, network_depth=4, network_width=32,
                       last_tanh=True),
    "SIREN_raw3": dict(network_input_size=3, network_output_size=2, network_depth=3, network_width=32),
    "FFN": dict(network_input_size=16, network_output_size=2, network_depth=4, network_width=32),
    "WIRE": dict(network_input_size=3, network_output_size=2, network_depth=2, network_width=46,
                 first_omega_0=30, hidden_omega_0=30, scale=15),
    "WIRE2D": dict(network_input_size=3, network_output_size=2, network_depth=2, network_width=32,
                   first_omega_0=20, hidden_omega_0=20, scale=10),
    "WIRE2D_tanh": dict(network_input_size=3, network_output_size=2, network_depth=2, network_width=32,
                        first_omega_0=20, hidden_omega_0=20, scale=10, last_tanh=True),
    "Fourier": dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32),
    "Gabor": dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32),
    "KGabor": dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32),
    "MultiscaleKFourier": dict(network_input_size=16, network_output_size=2, network_depth=8, network_width=32),
    "BoundedFourier": dict(network_input_size=16, network_output_size=2, network_depth=8, network_width=32),
}

FULL = {
    "SIREN": dict(network_input_size=512, network_output_size=2, network_depth=5, network_width=256, last_tanh=True),
    "SIREN4": dict(network_input_size=512, network_output_size=2, network_depth=4, network_width=256),
    "FFN": dict(network_input_size=512, network_output_size=2, network_depth=4, network_width=256),
    "WIRE": dict(network_input_size=3, network_output_size=2, network_depth=4, network_width=256,
                 first_omega_0=30, hidden_omega_0=30, scale=15),
    "WIRE2D": dict(network_input_size=3, network_output_size=2, network_depth=2, network_width=256,
                   first_omega_0=20, hidden_omega_0=20, scale=10),
    "Fourier": dict(network_input_size=512, network_output_size=2, network_depth=4, network_width=256),
    "Gabor": dict(network_input_size=512, network_output_size=2, network_depth=2, network_width=128),
    "MultiscaleKFourier": dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512),
    "BoundedFourier": dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=256),
}


def model_kind(name):
    return {"SIREN_tanh": "SIREN", "SIREN_raw3": "SIREN", "SIREN4": "SIREN", "WIRE2D_tanh": "WIRE2D"}.get(name, name)


def build(name, net, seed, enc=None):
    torch.manual_seed(seed)
    encoder = None
    if enc is not None:
        encoder = Positional_Encoder(enc, device="cpu")  # encoder first (train.py:52), then the model
    model = quiet(CTORS[model_kind(name)], net)
    return encoder, model


def call_model(name, model, x, dist):
    kind = model_kind(name)
    if kind == "KGabor":
        return model(x, dist)
    if kind in ("MultiscaleKFourier", "BoundedFourier"):
        return model(coords=x, dist_to_center=dist)
    return model(x)


def init_hashes():
    out = {}
    enc = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
    for name, net in FULL.items():
        seed = 0
        use_enc = enc if net["network_input_size"] == 512 else None
        encoder, model = build(name, net, seed, use_enc)
        ent = {"seed": seed, "net": net, "encoder": use_enc, "model": model_kind(name),
               "sha256": {k: sha(v) for k, v in model.state_dict().items()},
               "shapes": {k: list(v.shape) for k, v in model.state_dict().items()},
               "dtypes": {k: str(v.dtype) for k, v in model.state_dict().items()}}
        if encoder is not None:
            ent["enc_sha256"] = sha(encoder.B)
        ent["n_params"] = int(sum(p.numel() for p in model.parameters()))
        out[name] = ent
    with open(os.path.join(OUT, "init_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def model_vectors():
    """Per tiny model: state_dict, input, forward, 0.5*MSE loss, grads, params after 1 & 3 Adam steps."""
    meta = {}
    for name, net in TINY.items():
        seed = 7
        enc = None
        if net["network_input_size"] == 16:
            enc = dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)
        encoder, model = build(name, net, seed, enc)
        g = torch.Generator().manual_seed(123)
        coords = torch.rand(64, 3, generator=g) * 2 - 1
        gt = torch.randn(64, 2, generator=g) * 0.3
        dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2)
        x = encoder.embedding(coords) if encoder is not None else coords
        arrs = OrderedDict()
        arrs["coords"], arrs["gt"], arrs["x"] = npy(coords), npy(gt), npy(x)
        if encoder is not None:
            arrs["enc_B"] = npy(encoder.B)
        for k, v in model.state_dict().items():
            arrs["sd/" + k] = npy(v)
        multi = model_kind(name) in ("MultiscaleKFourier", "BoundedFourier")
        for wd_tag, wd in (("wd0", 0.0), ("wd1", 1e-2)):
            encoder, model = build(name, net, seed, enc)
            optim = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=wd)
            for step in range(1, 4):
                out = call_model(name, model, x, dist)
                optim.zero_grad()
                if multi:
                    loss = sum(0.5 * torch.nn.functional.mse_loss(o, gt) for o in out)
                else:
                    loss = 0.5 * torch.nn.functional.mse_loss(out, gt)
                loss.backward()
                if step == 1 and wd_tag == "wd0":
                    if multi:
                        for i, o in enumerate(out):
                            arrs[f"out/{i}"] = npy(o)
                    else:
                        arrs["out"] = npy(out)
                    arrs["loss"] = npy(loss)
                    for k, p in model.named_parameters():
                        if p.grad is not None:
                            arrs["grad/" + k] = npy(p.grad)
                optim.step()
                if step in (1, 3):
                    for k, v in model.state_dict().items():
                        arrs[f"{wd_tag}/step{step}/" + k] = npy(v)
        np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **arrs)
        meta[name] = {"seed": seed, "net": net, "encoder": enc, "model": model_kind(name),
                      "lr": 1e-3, "betas": [0.9, 0.999], "wd1": 1e-2,
                      "bounds8": BOUNDS8 if model_kind(name) == "BoundedFourier" else None,
                      "complex_keys": [k for k, v in model.state_dict().items() if v.is_complex()]}
    with open(os.path.join(OUT, "model_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def loss_vectors():
    g = torch.Generator().manual_seed(99)
    B = 96
    out = (torch.randn(B, 2, generator=g) * 0.2).requires_grad_(True)
    gt = torch.randn(B, 2, generator=g) * 0.2
    kc = torch.rand(B, 3, generator=g) * 2 - 1
    opts = dict(hdr_eps=1e-3, hdr_ff_sigma=2, hdr_ff_factor=0.5, min_sample=3000)
    arrs = {"out": npy(out), "gt": npy(gt), "kcoords": npy(kc)}

    def rec(tag, loss):
        (gr,) = torch.autograd.grad(loss, out)
        arrs[tag + "/loss"], arrs[tag + "/grad"] = npy(loss), npy(gr)

    l, reg = HDRLoss_FF(opts)(out, gt, kc)
    arrs["hdr/reg"] = npy(reg)
    rec("hdr", l)
    # masked variant of SURVEY A.4 #17: masked outputs [Bs,2] with the unmasked kcoords [B,3]
    mask = torch.rand(B, generator=g) < 0.4
    arrs["mask"] = mask.numpy()
    l, _ = HDRLoss_FF(opts)(out[mask], gt[mask], kc)
    rec("hdr_masked", l)
    rec("tanh", TanhL2Loss()(out, gt, kc)[0])
    rec("logspace", LogSpaceLoss(opts)(out, gt))
    rec("l2", 0.5 * torch.nn.MSELoss()(out, gt))
    rec("l1", 0.5 * torch.nn.L1Loss()(out, gt))
    pos = (torch.rand(B, 2, generator=g) + 0.1).requires_grad_(True)
    post = torch.rand(B, 2, generator=g) + 0.1
    lm = MSLELoss()(pos, post)
    arrs["msle/x"], arrs["msle/y"], arrs["msle/loss"] = npy(pos), npy(post), npy(lm)
    arrs["msle/grad"] = npy(torch.autograd.grad(lm, pos)[0])
    # TV on [8,6,2]
    img = torch.randn(8, 6, 2, generator=g).requires_grad_(True)
    lt = tv_loss(img)
    arrs["tv/img"], arrs["tv/loss"] = npy(img), npy(lt)
    arrs["tv/grad"] = npy(torch.autograd.grad(lt, img)[0])
    # Consistency: dist [B] and [B,1]
    outs = [(torch.randn(B, 2, generator=g) * 0.2).requires_grad_(True) for _ in range(4)]
    dist = torch.sqrt(kc[:, 1] ** 2 + kc[:, 2] ** 2)
    for i, o in enumerate(outs):
        arrs[f"cons/out{i}"] = npy(o)
    arrs["cons/dist"] = npy(dist)
    for tag, d in (("cons_flat", dist), ("cons_col", dist[:, None])):
        lc = ConsistencyLoss(BOUNDS4)(outs, d)
        arrs[tag + "/loss"] = npy(lc)
        grs = torch.autograd.grad(lc, outs, allow_unused=True)
        for i, gr in enumerate(grs):
            arrs[f"{tag}/grad{i}"] = npy(gr if gr is not None else torch.zeros_like(outs[i]))
    # Regularisers on a small parameter list
    ps = [torch.randn(5, 3, generator=g).requires_grad_(True), torch.randn(5, generator=g).requires_grad_(True)]
    arrs["reg/p0"], arrs["reg/p1"] = npy(ps[0]), npy(ps[1])
    for tag, R in (("reg_l1", Regularization_L1), ("reg_l2", Regularization_L2)):
        lr_ = R(reg_strength=0.003)(ps)
        arrs[tag + "/loss"] = npy(lr_)
        grs = torch.autograd.grad(lr_, ps)
        arrs[tag + "/grad0"], arrs[tag + "/grad1"] = npy(grs[0]), npy(grs[1])
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **arrs)
    with open(os.path.join(OUT, "losses_meta.json"), "w") as f:
        json.dump({"opts": opts, "bounds4": BOUNDS4, "reg_strength": 0.003}, f, indent=1)


def synth_kspace(C, H, W, seed):
    """Small smooth synthetic multi-coil k-space (data only; not the product's generator)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    img = np.zeros((H, W))
    for _ in range(4):
        cx, cy, a, b, v = rng.uniform(-.5, .5), rng.uniform(-.5, .5), rng.uniform(.2, .6), rng.uniform(.2, .6), rng.uniform(.3, 1)
        img += v * (((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 < 1)
    coils = []
    for c in range(C):
        ang = 2 * np.pi * c / C
        sens = np.exp(-((xx - .8 * np.cos(ang)) ** 2 + (yy - .8 * np.sin(ang)) ** 2)) * np.exp(1j * (xx * np.cos(ang) + yy * np.sin(ang)))
        im = img * sens
        k = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(im), norm="ortho"))
        coils.append(k)
    k = np.stack(coils)
    k = np.stack([k.real, k.imag], -1).astype(np.float32)
    k = k / np.abs(k).max()
    return torch.from_numpy(k)


def trajectory():
    """Short training trajectories driven by the loop of train.py:158-192 (re-stated here because
    train.py itself needs fastmri/h5py/tensorboard) around the IMPORTED reference model/loss classes
    and stock torch.optim.Adam + LambdaLR."""
    from torch.optim.lr_scheduler import LambdaLR
    C, H, W = 2, 24, 20
    k = synth_kspace(C, H, W, 5)
    image = k.reshape(C * H * W, 2)
    Z, Y, X = torch.meshgrid(torch.linspace(-1, 1, C), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    coords = torch.hstack((Z.reshape(-1, 1), Y.reshape(-1, 1), X.reshape(-1, 1)))
    cases = {
        "SIREN_L2": dict(model="SIREN", loss="L2", lr=1e-4, batch_size=300, max_epoch=3, weight_decay=0.0,
                         beta1=0.9, beta2=0.999,
                         net=dict(network_input_size=16, network_output_size=2, network_depth=4, network_width=32,
                                  last_tanh=True),
                         encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)),
        "SIREN_L2_reg": dict(model="SIREN", loss="L2", lr=1e-4, batch_size=300, max_epoch=3, weight_decay=1e-3,
                             beta1=0.9, beta2=0.999, regularization=dict(type="L1", strenght=1e-5),
                             net=dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32),
                             encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)),
        "SIREN_regL2": dict(model="SIREN", loss="L2", lr=1e-4, batch_size=300, max_epoch=3, weight_decay=0.0,
                            beta1=0.9, beta2=0.999, regularization=dict(type="L2", strenght=1e-4),
                            net=dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32),
                            encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)),
        "WIRE_HDR": dict(model="WIRE", loss="HDR", lr=1e-4, batch_size=240, max_epoch=3, weight_decay=0.0,
                         beta1=0.9, beta2=0.999,
                         loss_opts=dict(hdr_eps=1e-3, hdr_ff_sigma=2, hdr_ff_factor=0.5),
                         net=dict(network_input_size=3, network_output_size=2, network_depth=2, network_width=46,
                                  first_omega_0=30, hidden_omega_0=30, scale=15),
                         encoder=dict(embedding="none", scale=0, embedding_size=0, coordinates_size=3)),
        "SIREN_LSL": dict(model="SIREN", loss="LSL", lr=1e-4, batch_size=480, max_epoch=3, weight_decay=0.0,
                          beta1=0.9, beta2=0.999,
                          loss_opts=dict(hdr_eps=3e-3, hdr_ff_sigma=2, hdr_ff_factor=0.5, min_sample=40),
                          net=dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32,
                                   last_tanh=True),
                          encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)),
        "Fourier_tanh": dict(model="Fourier", loss="tanh", lr=1e-3, batch_size=480, max_epoch=4, weight_decay=0.0,
                             beta1=0.9, beta2=0.999,
                             net=dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32),
                             encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)),
    }
    arrs = {"image": npy(image), "coords": npy(coords)}
    CKPT_STEP = 5
    meta = {"shape": [C, H, W], "cases": cases, "seed": 3, "steps": 12, "checkpoint_step": CKPT_STEP}
    for tag, cfg in cases.items():
        torch.manual_seed(3)
        encoder = Positional_Encoder(cfg["encoder"], device="cpu")
        model = quiet(CTORS[cfg["model"]], cfg["net"])
        optim = torch.optim.Adam(model.parameters(), lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]),
                                 weight_decay=cfg["weight_decay"])
        if cfg["loss"] == "L2":
            loss_fn = torch.nn.MSELoss()
        elif cfg["loss"] == "HDR":
            loss_fn = HDRLoss_FF(cfg["loss_opts"])
        elif cfg["loss"] == "tanh":
            loss_fn = TanhL2Loss()
        elif cfg["loss"] == "LSL":  # CenterLoss: its torch.randperm draws continue the generator seeded above
            loss_fn = CenterLoss(cfg["loss_opts"])
        reg = None
        if cfg.get("regularization", {}).get("type") == "L1":
            reg = Regularization_L1(reg_strength=cfg["regularization"]["strenght"])
        elif cfg.get("regularization", {}).get("type") == "L2":
            reg = Regularization_L2(reg_strength=cfg["regularization"]["strenght"])
        sched = LambdaLR(optim, lambda x: 0.2 ** min(x / cfg["max_epoch"], 1))
        bs = cfg["batch_size"]
        losses, step = [], 0
        for epoch in range(cfg["max_epoch"]):
            for lo in range(0, coords.shape[0], bs):
                if step >= meta["steps"]:
                    break
                kc = coords[lo:lo + bs]
                gt = image[lo:lo + bs]
                out = model(encoder.embedding(kc))
                # WIRE returns ``output.real`` (non-contiguous); the reference's HDRLoss_FF then raises in
                # view_as_complex (networks.py:258 vs losses.py:245).  The golden is taken on the contiguous
                # copy, i.e. the only arithmetic the combination can mean.
                out = out.contiguous()
                optim.zero_grad()
                if cfg["loss"] in ("HDR", "tanh", "LSL"):
                    loss, _ = loss_fn(out, gt, kc)
                else:
                    loss = 0.5 * loss_fn(out, gt)
                if reg is not None:
                    loss = loss + reg(model.parameters())
                loss.backward()
                optim.step()
                losses.append(float(loss.detach()))
                step += 1
                if tag == "SIREN_L2" and step == CKPT_STEP:
                    # a checkpoint exactly as the reference writes it (train.py:244-250): reference SIREN state_dict,
                    # encoder.B, stock torch.optim.Adam state -- tensors only, loaded by tests/test_gpu_configs.py
                    torch.save({"net": model.state_dict(), "enc": encoder.B, "opt": optim.state_dict()},
                               os.path.join(OUT, "ref_checkpoint_SIREN_L2_step%d.pt" % CKPT_STEP))
            sched.step()
        arrs[tag + "/losses"] = np.array(losses, dtype=np.float64)
        with torch.no_grad():
            arrs[tag + "/final_out"] = npy(model(encoder.embedding(coords)))
        for kname, v in model.state_dict().items():
            arrs[f"{tag}/final_sd/{kname}"] = npy(v)
    np.savez_compressed(os.path.join(OUT, "trajectory.npz"), **arrs)
    with open(os.path.join(OUT, "trajectory_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def center_vectors():
    """CenterLoss.forward (losses.py:141-201) value and gradient; the pairs are whatever torch.randperm draws after
    torch.manual_seed(SEED) -- the oracle and the engine's host side draw the same way."""
    g = torch.Generator().manual_seed(77)
    B, SEED = 800, 4321
    out = (torch.randn(B, 2, generator=g) * 0.2).requires_grad_(True)
    gt = torch.randn(B, 2, generator=g) * 0.2
    kc = torch.rand(B, 3, generator=g) * 2 - 1
    arrs = {"out": npy(out), "gt": npy(gt), "kcoords": npy(kc), "seed": np.array(SEED)}
    for tag, ms in (("ms50", 50), ("ms3000", 3000)):
        opts = dict(hdr_eps=1e-3, hdr_ff_sigma=2, hdr_ff_factor=0.5, min_sample=ms)
        torch.manual_seed(SEED)
        loss, _ = CenterLoss(opts)(out, gt, kc)
        (gr,) = torch.autograd.grad(loss, out)
        arrs[tag + "/loss"], arrs[tag + "/grad"] = npy(loss), npy(gr)
    np.savez_compressed(os.path.join(OUT, "center.npz"), **arrs)


def multiscale_trajectory():
    """train_kspace_multiscale.py:164-195 re-stated around the imported MultiscaleKFourier /
    MultiscaleBoundedFourier + LogSpaceLoss + ConsistencyLoss (no undersampling, no TV)."""
    from torch.optim.lr_scheduler import LambdaLR
    C, H, W = 2, 24, 20
    k = synth_kspace(C, H, W, 6)
    image = k.reshape(C * H * W, 2)
    Z, Y, X = torch.meshgrid(torch.linspace(-1, 1, C), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    coords = torch.hstack((Z.reshape(-1, 1), Y.reshape(-1, 1), X.reshape(-1, 1)))
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2)
    radii = [0.0, 0.2, 0.45, 0.8, 5.0]
    pairs = [(radii[0], radii[i + 1]) for i in range(4)]
    pairs_model = [p for p in pairs for _ in range(2)]
    arrs = {"image": npy(image), "coords": npy(coords), "dist": npy(dist)}
    cases = {}
    for tag, mname in (("MS_LSL", "MultiscaleKFourier"), ("Bounded_L2", "BoundedFourier")):
        cfg = dict(model=mname, loss="LSL" if tag == "MS_LSL" else "L2", lr=3e-4, batch_size=400, max_epoch=3,
                   weight_decay=0.0, beta1=0.9, beta2=0.999,
                   loss_opts=dict(hdr_eps=3e-3, hdr_ff_sigma=2, hdr_ff_factor=0.5),
                   net=dict(network_input_size=16, network_output_size=2, network_depth=8, network_width=32),
                   encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3))
        cases[tag] = cfg
        torch.manual_seed(4)
        encoder = Positional_Encoder(cfg["encoder"], device="cpu")
        if mname == "BoundedFourier":
            model = quiet(MultiscaleBoundedFourier, cfg["net"], boundaries=pairs_model)
        else:
            model = quiet(MultiscaleKFourier, cfg["net"])
        optim = torch.optim.Adam(model.parameters(), lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=0.0)
        loss_fn = LogSpaceLoss(cfg["loss_opts"]) if cfg["loss"] == "LSL" else torch.nn.MSELoss()
        loss_cons = ConsistencyLoss(pairs)
        sched = LambdaLR(optim, lambda x: 0.2 ** min(x / cfg["max_epoch"], 1))
        losses, step, bs = [], 0, cfg["batch_size"]
        for epoch in range(cfg["max_epoch"]):
            for lo in range(0, coords.shape[0], bs):
                if step >= 8:
                    break
                kc, gt, d = coords[lo:lo + bs], image[lo:lo + bs], dist[lo:lo + bs]
                outs = model(coords=encoder.embedding(kc), dist_to_center=d)
                optim.zero_grad()
                loss = 0.1 * loss_cons(outs, d)
                for o in outs:
                    loss = loss + 0.5 * loss_fn(o, gt)
                loss.backward()
                optim.step()
                losses.append(float(loss.detach()))
                step += 1
                if tag == "SIREN_L2" and step == CKPT_STEP:
                    # a checkpoint exactly as the reference writes it (train.py:244-250): reference SIREN state_dict,
                    # encoder.B, stock torch.optim.Adam state -- tensors only, loaded by tests/test_gpu_configs.py
                    torch.save({"net": model.state_dict(), "enc": encoder.B, "opt": optim.state_dict()},
                               os.path.join(OUT, "ref_checkpoint_SIREN_L2_step%d.pt" % CKPT_STEP))
            sched.step()
        arrs[tag + "/losses"] = np.array(losses, dtype=np.float64)
        with torch.no_grad():
            arrs[tag + "/final_out"] = npy(model(coords=encoder.embedding(coords), dist_to_center=dist)[-1])
    # per-coil batches + grid undersampling + TV on the last head (train_kspace_multiscale.py:164-195 with
    # use_tv and len(mask_coords) != 0): TV sees train_output[-1] of ALL rows, the consistency term all rows,
    # the pointwise terms the sampled rows
    mask2d = torch.zeros(H, W, dtype=torch.bool)
    mask2d[::2, ::3] = True
    mask = mask2d.reshape(1, -1).expand(C, -1).reshape(-1)
    masked_image = image * mask[:, None]
    arrs["mask"] = mask.numpy()
    cfg = dict(model="MultiscaleKFourier", loss="LSL", lr=3e-4, batch_size=1, max_epoch=4, weight_decay=0.0, beta1=0.9,
               beta2=0.999, per_coil=True, use_tv=True, undersampling="grid-2*3",
               loss_opts=dict(hdr_eps=3e-3, hdr_ff_sigma=2, hdr_ff_factor=0.5),
               net=dict(network_input_size=16, network_output_size=2, network_depth=8, network_width=32),
               encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3))
    cases["MS_percoil_tv"] = cfg
    torch.manual_seed(4)
    encoder = Positional_Encoder(cfg["encoder"], device="cpu")
    model = quiet(MultiscaleKFourier, cfg["net"])
    optim = torch.optim.Adam(model.parameters(), lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=0.0)
    loss_fn = LogSpaceLoss(cfg["loss_opts"])
    loss_cons = ConsistencyLoss(pairs)
    sched = LambdaLR(optim, lambda x: 0.2 ** min(x / cfg["max_epoch"], 1))
    losses = []
    for epoch in range(cfg["max_epoch"]):
        for c in range(C):
            sl = slice(c * H * W, (c + 1) * H * W)
            kc, gt, d, m = coords[sl], masked_image[sl], dist[sl], mask[sl]
            outs = model(coords=encoder.embedding(kc), dist_to_center=d)
            optim.zero_grad()
            loss = tv_loss(outs[-1].view((H, W, 2)))
            gt_m = gt[m]
            loss = loss + 0.1 * loss_cons(outs, d)
            for o in outs:
                loss = loss + 0.5 * loss_fn(o[m], gt_m)
            loss.backward()
            optim.step()
            losses.append(float(loss.detach()))
        sched.step()
    arrs["MS_percoil_tv/losses"] = np.array(losses, dtype=np.float64)
    with torch.no_grad():
        arrs["MS_percoil_tv/final_out"] = npy(model(coords=encoder.embedding(coords), dist_to_center=dist)[-1])
    np.savez_compressed(os.path.join(OUT, "trajectory_ms.npz"), **arrs)
    with open(os.path.join(OUT, "trajectory_ms_meta.json"), "w") as f:
        json.dump({"shape": [C, H, W], "cases": cases, "seed": 4, "steps": 8, "radii": radii}, f, indent=1, sort_keys=True)


def undersampling_vectors():
    """Masks from the reference's Undersampler (undersampling/undersampler.py) and a per-coil + TV
    trajectory (train.py:158-192 with len(mask_coords) != 0 and use_tv)."""
    from torch.optim.lr_scheduler import LambdaLR
    from undersampling.undersampler import Undersampler
    cwd = os.getcwd()
    os.chdir("/tmp")  # the reference saves undersampling_mask.png into the cwd
    try:
        arrs = {}
        # radial, BASELINE shape, acc 4: the reference draws t from an UNSEEDED RandomState
        # (undersampler.py:115,123); pin it by handing it a seeded one
        real_rs = np.random.RandomState
        np.random.RandomState = lambda *a, **k: real_rs(11)
        try:
            u = Undersampler("radial")
            quiet(u.create_mask_for_radial_based_undersampling, (15, 640, 368, 2), 4, False)
            m = u._Undersampler__mask_image
            arrs["radial_640x368_acc4"] = np.packbits(m.numpy().astype(np.uint8))
            t = real_rs(11).randint(low=0, high=1e4, size=1, dtype=int).item()
            u2 = Undersampler("radial")
            quiet(u2.create_mask_for_radial_based_undersampling, (3, 65, 48, 2), 2, False)  # odd side: pad path
            arrs["radial_65x48_acc2"] = np.packbits(u2._Undersampler__mask_image.numpy().astype(np.uint8))
        finally:
            np.random.RandomState = real_rs
        # grid 3x2 through apply(): masked data, grid and the [N,3] bool coordinate mask
        C, H, W = 2, 24, 20
        k = synth_kspace(C, H, W, 8)
        masked, grid, grid_mask = quiet(Undersampler("grid").apply, k, [3, 2])
        arrs["grid_full"], arrs["grid_masked"], arrs["grid_coords"] = npy(k), npy(masked), npy(grid)
        arrs["grid_mask"] = grid_mask.numpy()
        # per-coil + TV trajectory: one step = one coil (MRICoilWrapperDataset, nerp_datasets.py:397-441)
        cfg = dict(model="SIREN", loss="L2", lr=2e-4, batch_size=H * W, max_epoch=4, weight_decay=0.0, beta1=0.9,
                   beta2=0.999, per_coil=True, use_tv=True, undersampling="grid-3*2",
                   net=dict(network_input_size=16, network_output_size=2, network_depth=4, network_width=32),
                   encoder=dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3))
        torch.manual_seed(5)
        encoder = Positional_Encoder(cfg["encoder"], device="cpu")
        model = quiet(SIREN, cfg["net"])
        optim = torch.optim.Adam(model.parameters(), lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=0.0)
        sched = LambdaLR(optim, lambda x: 0.2 ** min(x / cfg["max_epoch"], 1))
        image = masked.reshape(C * H * W, 2)
        losses = []
        for epoch in range(cfg["max_epoch"]):
            for c in range(C):
                sl = slice(c * H * W, (c + 1) * H * W)
                coords, gt, mask_coords = grid[sl], image[sl], grid_mask[sl]
                out = model(encoder.embedding(coords))
                optim.zero_grad()
                loss = tv_loss(out.view((H, W, 2)))
                out_m, gt_m = out[mask_coords[:, 0]], gt[mask_coords[:, 0]]
                loss = loss + 0.5 * torch.nn.functional.mse_loss(out_m, gt_m)
                loss.backward()
                optim.step()
                losses.append(float(loss.detach()))
            sched.step()
        arrs["percoil_tv/losses"] = np.array(losses, dtype=np.float64)
        with torch.no_grad():
            arrs["percoil_tv/final_out"] = npy(model(encoder.embedding(grid)))
        np.savez_compressed(os.path.join(OUT, "undersampling.npz"), **arrs)
        with open(os.path.join(OUT, "undersampling_meta.json"), "w") as f:
            json.dump({"radial_seed": 11, "radial_t": t, "shape": [C, H, W], "grid": [3, 2], "config": cfg, "seed": 5},
                      f, indent=1, sort_keys=True)
    finally:
        os.chdir(cwd)


def extra_trajectories():
    """Trajectories of loop branches the first rounds refused: Regularization_L1 / _L2 (regularization.py:21-36) on the
    complex64 parameters of WIRE / WIRE2D (train.py:185-187), and tv_loss (train.py:172-175) with filter networks on per-coil
    batches.  Same re-stated loop as trajectory() / undersampling_vectors() around the imported reference classes."""
    from torch.optim.lr_scheduler import LambdaLR
    from undersampling.undersampler import Undersampler
    C, H, W = 2, 24, 20
    k = synth_kspace(C, H, W, 9)
    Z, Y, X = torch.meshgrid(torch.linspace(-1, 1, C), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    coords_full = torch.hstack((Z.reshape(-1, 1), Y.reshape(-1, 1), X.reshape(-1, 1)))
    cwd = os.getcwd()
    os.chdir("/tmp")  # the reference saves undersampling_mask.png into the cwd
    try:
        masked, grid, grid_mask = quiet(Undersampler("grid").apply, k, [3, 2])
    finally:
        os.chdir(cwd)
    base = dict(loss="L2", lr=2e-4, max_epoch=3, weight_decay=0.0, beta1=0.9, beta2=0.999)
    none_enc = dict(embedding="none", scale=0, embedding_size=0, coordinates_size=3)
    gauss_enc = dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)
    cases = {
        "WIRE_regL1": dict(base, model="WIRE", batch_size=240, regularization=dict(type="L1", strenght=1e-5),
                           net=TINY["WIRE"], encoder=none_enc),
        "WIRE_regL2": dict(base, model="WIRE", batch_size=240, regularization=dict(type="L2", strenght=1e-4),
                           net=TINY["WIRE"], encoder=none_enc),
        "WIRE2D_regL2": dict(base, model="WIRE2D", batch_size=240, regularization=dict(type="L2", strenght=1e-4),
                             weight_decay=1e-3, net=TINY["WIRE2D_tanh"], encoder=none_enc),
        "Fourier_percoil_tv": dict(base, model="Fourier", batch_size=H * W, per_coil=True, use_tv=True,
                                   undersampling="grid-3*2", net=TINY["Fourier"], encoder=gauss_enc),
        "Gabor_percoil_tv": dict(base, model="Gabor", batch_size=H * W, per_coil=True, use_tv=True,
                                 undersampling="grid-3*2", net=TINY["Gabor"], encoder=gauss_enc),
    }
    arrs = {"full": npy(k), "masked": npy(masked), "coords": npy(grid), "mask": grid_mask.numpy()}
    assert torch.equal(grid.reshape(-1, 3), coords_full)
    meta = {"shape": [C, H, W], "grid": [3, 2], "cases": cases, "seed": 4, "steps": 8}
    for tag, cfg in cases.items():
        torch.manual_seed(meta["seed"])
        encoder = Positional_Encoder(cfg["encoder"], device="cpu")
        model = quiet(CTORS[cfg["model"]], cfg["net"])
        optim = torch.optim.Adam(model.parameters(), lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]),
                                 weight_decay=cfg["weight_decay"])
        sched = LambdaLR(optim, lambda x: 0.2 ** min(x / cfg["max_epoch"], 1))
        reg = None
        if cfg.get("regularization", {}).get("type") == "L1":
            reg = Regularization_L1(reg_strength=cfg["regularization"]["strenght"])
        elif cfg.get("regularization", {}).get("type") == "L2":
            reg = Regularization_L2(reg_strength=cfg["regularization"]["strenght"])
        tv = cfg.get("use_tv", False)
        image = (masked if tv else k).reshape(C * H * W, 2)
        coords = grid.reshape(-1, 3)
        bs = cfg["batch_size"]
        losses, step = [], 0
        for epoch in range(cfg["max_epoch"]):
            for lo in range(0, coords.shape[0], bs):
                if step >= meta["steps"]:
                    break
                kc, gt = coords[lo:lo + bs], image[lo:lo + bs]
                out = model(encoder.embedding(kc))
                optim.zero_grad()
                loss = 0
                if tv:  # train.py:172-177
                    loss = loss + tv_loss(out.view((H, W, 2)))
                    m = grid_mask.reshape(-1, 3)[lo:lo + bs, 0]
                    out, gt = out[m], gt[m]
                loss = loss + 0.5 * torch.nn.functional.mse_loss(out, gt)
                if reg is not None:
                    loss = loss + reg(model.parameters())
                loss.backward()
                optim.step()
                losses.append(float(loss.detach()))
                step += 1
            sched.step()
        arrs[tag + "/losses"] = np.array(losses, dtype=np.float64)
        with torch.no_grad():
            arrs[tag + "/final_out"] = npy(model(encoder.embedding(coords)).contiguous())
        for kname, v in model.state_dict().items():
            arrs[f"{tag}/final_sd/{kname}"] = npy(v)
    np.savez_compressed(os.path.join(OUT, "trajectory_extra.npz"), **arrs)
    with open(os.path.join(OUT, "trajectory_extra_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def clustering_vectors():
    """Ring partition (clustering.py:19-135) on a small synthetic k-space.  clustering.py imports
    models.utils at module scope only for its __main__ block (get_config / get_data_loader); that module drags in
    h5py / torchvision, which are absent here, so an empty stand-in is registered for it -- the two functions
    exercised below do not touch it."""
    import matplotlib
    matplotlib.use("Agg")
    mu = types.ModuleType("models.utils")
    mu.get_config = mu.get_data_loader = None
    sys.modules.setdefault("models.utils", mu)
    import clustering as ref_clustering
    arrs, meta = {}, {"cases": {}}
    for tag, (C, H, W, steps, parts) in {"a": (3, 48, 40, 40, 4), "b": (2, 64, 64, 24, 3)}.items():
        k = synth_kspace(C, H, W, 21 + C)
        grid = torch.stack(torch.meshgrid(torch.linspace(-1, 1, C), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W),
                                          indexing="ij"), dim=-1)
        labels, radii = quiet(ref_clustering.partition_kspace, None, k, grid, False, steps, parts)
        stats, radii2 = quiet(ref_clustering.partition_and_stats, None, k, grid, False, steps, parts, "max")
        assert np.array_equal(radii, radii2)
        arrs[f"{tag}/kspace"], arrs[f"{tag}/coords"] = npy(k), npy(grid)
        arrs[f"{tag}/labels"], arrs[f"{tag}/radii"], arrs[f"{tag}/stats"] = np.asarray(labels), np.asarray(radii), npy(stats)
        meta["cases"][tag] = {"shape": [C, H, W], "no_steps": steps, "no_parts": parts}
    np.savez_compressed(os.path.join(OUT, "clustering.npz"), **arrs)
    with open(os.path.join(OUT, "clustering_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def ingest_vectors():
    """The reference's OWN ingest arithmetic (not third-party): MRIDataset.__normalize_kspace for its seven schemes
    (nerp_datasets.py:108-143), retrieve_size on an ISMRMRD header (:151-170), and complex_center_crop /
    normalize_image / gaussian_filter_2d / create_coords of data/utils.py:19-28,65-108.  nerp_datasets.py imports h5py and
    fastmri.data.transforms at module scope (both absent here, neither touched by these functions): empty stand-ins are
    registered for them, fastmri.complex_abs is the in-memory formula this script already uses.  What stays unpinned is
    what fastmri itself computes (ifft2c / fft2c)."""
    import matplotlib
    matplotlib.use("Agg")
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    fmd = types.ModuleType("fastmri.data")
    fmd.transforms = types.ModuleType("fastmri.data.transforms")
    sys.modules.setdefault("fastmri.data", fmd)
    sys.modules.setdefault("fastmri.data.transforms", fmd.transforms)
    from data.nerp_datasets import MRIDataset
    from data import utils as ref_utils
    norm = MRIDataset._MRIDataset__normalize_kspace
    g = torch.Generator().manual_seed(77)
    arrs, meta = {}, {"normalizations": ["abs_max", "max", "gaussian_blur", "max_std", "tonemap", "coil", "stand", "none"]}
    k = torch.randn(3, 20, 14, 2, generator=g) * torch.tensor([1.0, 0.2, 3.0])[:, None, None, None]
    k[1, 10, 7] = torch.tensor([9.0, -4.0])  # a k-space centre: the maxima differ between coils and components
    arrs["kspace"] = npy(k)
    for n in meta["normalizations"]:
        arrs[f"norm/{n}"] = npy(quiet(norm, k.clone(), n))
    img = torch.randn(2, 18, 11, 2, generator=g)
    arrs["image"] = npy(img)
    arrs["normalize_image"] = npy(ref_utils.normalize_image(img))
    crops = {"inside": (10, 6), "wider_than_w": (12, 20), "full": (18, 11)}
    for tag, shp in crops.items():
        arrs[f"crop/{tag}"] = npy(ref_utils.complex_center_crop(img, shp))
    meta["crops"] = {t: list(v) for t, v in crops.items()}
    x = torch.randn(2, 1, 9, 7, generator=g)
    arrs["blur_in"], arrs["blur_out"] = npy(x), npy(ref_utils.gaussian_filter_2d(x, 0.1))
    arrs["blur_out_sigma1"] = npy(ref_utils.gaussian_filter_2d(x, 1.0))
    arrs["coords_3_5_4"] = npy(ref_utils.create_coords(3, 5, 4))
    header = ('<?xml version="1.0" encoding="utf-8"?><ismrmrdHeader xmlns="http://www.ismrm.org/ISMRMRD"><encoding>'
              '<encodedSpace><matrixSize><x>640</x><y>372</y><z>1</z></matrixSize></encodedSpace>'
              '<reconSpace><matrixSize><x>320</x><y>322</y><z>1</z></matrixSize></reconSpace>'
              '<encodingLimits><kspace_encoding_step_1><minimum>0</minimum><maximum>367</maximum><center>184</center>'
              '</kspace_encoding_step_1></encodingLimits></encoding></ismrmrdHeader>')

    class _Field:  # what h5py hands back for file["ismrmrd_header"][()]
        def __init__(self, b):
            self.b = b

        def __getitem__(self, _):
            return self.b

    meta["header"] = header
    meta["recon_size"] = list(MRIDataset.retrieve_size({"ismrmrd_header": _Field(header.encode())}))
    np.savez_compressed(os.path.join(OUT, "ingest.npz"), **arrs)
    with open(os.path.join(OUT, "ingest_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    parts = dict(ingest=ingest_vectors, clustering=clustering_vectors, undersampling=undersampling_vectors, init=init_hashes,
                 models=model_vectors, losses=loss_vectors, center=center_vectors, trajectory=trajectory,
                 multiscale=multiscale_trajectory, extra=extra_trajectories)
    for name in (sys.argv[1:] or list(parts)):  # python tools/make_golden.py [part ...]; default: everything
        parts[name]()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, f"({tot / 1e6:.2f} MB)")
