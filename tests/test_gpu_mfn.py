"""Multiplicative filter networks (models/mfn.py: FourierNet, GaborNet, KGaborNet, MultiscaleKFourier,
MultiscaleBoundedFourier) on the HIP path."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402  (checker only)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLD, "model_meta.json")))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _classes():
    from inr_mi355x.mfn import FourierNet, GaborNet, KGaborNet, MultiscaleBoundedFourier, MultiscaleKFourier
    b8 = META["BoundedFourier"]["bounds8"]
    return {"Fourier": FourierNet, "MultiscaleKFourier": MultiscaleKFourier, "Gabor": GaborNet, "KGabor": KGaborNet,
            "BoundedFourier": lambda net: MultiscaleBoundedFourier(net, boundaries=b8)}


def _call_like_the_reference(name, model, x, dist):
    """train.py:163-169 / train_kspace_multiscale.py:169-170: the model receives encoder.embedding(coords)."""
    if name == "KGabor":
        return model(x, dist)
    if name in ("MultiscaleKFourier", "BoundedFourier"):
        return model(coords=x, dist_to_center=dist)
    return model(x)


@pytest.mark.parametrize("name", ["Fourier", "MultiscaleKFourier", "BoundedFourier", "Gabor", "KGabor"])
def test_mfn_tier1_golden(dev, name):
    """Drop-in class called EXACTLY as the reference calls it -- ``model(encoder.embedding(coords))`` on the encoded
    [B, network_input_size] input, no bind_encoder -- + stock torch.optim.Adam vs the reference's vectors; dead layers
    of the multiscale net keep grad None and are not stepped."""
    import inr_mi355x as M
    meta = META[name]
    arrs = _load(f"model_{name}.npz")
    coords, gt = _t(arrs["coords"]).to(dev), _t(arrs["gt"]).to(dev)
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
    gold_keys = [k[3:] for k in arrs if k.startswith("sd/")]
    for wd_tag, wd in (("wd0", 0.0), ("wd1", meta["wd1"])):
        torch.manual_seed(meta["seed"])
        enc = M.Positional_Encoder(meta["encoder"], device=dev)
        model = _classes()[name](meta["net"])
        sd = model.state_dict()
        assert list(sd.keys()) == gold_keys
        for k in gold_keys:
            if name in ("Gabor", "KGabor") and k.endswith("linear.weight") and k.startswith("filters"):
                # weight *= scale * sqrt(gamma) (mfn.py:112): torch's own CPU kernels round this 1 ulp apart on
                # different hosts (AVX512 Intel vs AMD), for the reference as well
                torch.testing.assert_close(sd[k], _t(arrs["sd/" + k]), rtol=3e-7, atol=0)
            else:
                assert torch.equal(sd[k], _t(arrs["sd/" + k])), k
        model.load_state_dict({k: _t(arrs["sd/" + k]) for k in gold_keys})  # pin the start to the vectors' own
        model = model.to(dev)
        optim = torch.optim.Adam(model.parameters(), lr=meta["lr"], betas=(0.9, 0.999), weight_decay=wd)
        for step in range(1, 4):
            out = _call_like_the_reference(name, model, enc.embedding(coords), dist)
            optim.zero_grad()
            outs = out if isinstance(out, list) else [out]
            loss = sum(0.5 * torch.nn.functional.mse_loss(o, gt) for o in outs)
            loss.backward()
            if step == 1 and wd_tag == "wd0":
                if isinstance(out, list):
                    for i, o in enumerate(out):
                        torch.testing.assert_close(o.detach().cpu(), _t(arrs[f"out/{i}"]), rtol=1e-5, atol=2e-6)
                else:
                    torch.testing.assert_close(out.detach().cpu(), _t(arrs["out"]), rtol=1e-5, atol=2e-6)
                torch.testing.assert_close(loss.detach().cpu(), _t(arrs["loss"]), rtol=1e-5, atol=0)
                gold_grad = {k[5:] for k in arrs if k.startswith("grad/")}
                for k, p in model.named_parameters():
                    if k in gold_grad:
                        ref = _t(arrs["grad/" + k])
                        assert rel_l2(p.grad.cpu(), ref) < 1e-5, (k, rel_l2(p.grad.cpu(), ref))
                    else:
                        assert p.grad is None, k
            optim.step()
            if step in (1, 3):
                for k, v in model.state_dict().items():
                    ref = _t(arrs[f"{wd_tag}/step{step}/" + k])
                    torch.testing.assert_close(v.cpu(), ref, rtol=1e-5, atol=2e-6,
                                               msg=lambda m: f"{wd_tag} step{step} {k}: {m}")


@pytest.mark.parametrize("name", ["Fourier", "MultiscaleKFourier", "BoundedFourier", "Gabor", "KGabor"])
def test_mfn_tier2_fused_golden(dev, name):
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    meta = META[name]
    arrs = _load(f"model_{name}.npz")
    coords, gt = _t(arrs["coords"]).to(dev), _t(arrs["gt"]).to(dev)
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
    for wd_tag, wd in (("wd0", 0.0), ("wd1", meta["wd1"])):
        torch.manual_seed(meta["seed"])
        enc = M.Positional_Encoder(meta["encoder"], device=dev)
        model = _classes()[name](meta["net"])
        model.load_state_dict({k[3:]: _t(v) for k, v in arrs.items() if k.startswith("sd/")})
        model = model.to(dev).bind_encoder(enc)
        eng = model._engine()
        for step in range(1, 4):
            loss = eng.train_step(coords, enc.B.contiguous(), gt, M.LossSpec(L.LOSS_L2_HALF), dist=dist)
            if step == 1 and wd_tag == "wd0":
                torch.testing.assert_close(loss.cpu(), _t(arrs["loss"]), rtol=1e-5, atol=0)
            eng.adam_step(meta["lr"], 0.9, 0.999, 1e-8, wd)
            if step in (1, 3):
                for k, v in model.state_dict().items():
                    ref = _t(arrs[f"{wd_tag}/step{step}/" + k])
                    torch.testing.assert_close(v.cpu(), ref, rtol=1e-5, atol=2e-6,
                                               msg=lambda m: f"{wd_tag} step{step} {k}: {m}")


def test_bounded_trajectory_golden(dev):
    from inr_mi355x.train_kspace_multiscale import MultiscaleTrainer
    arrs = _load("trajectory_ms.npz")
    meta = json.load(open(os.path.join(GOLD, "trajectory_ms_meta.json")))
    coords, image, dist = _t(arrs["coords"]), _t(arrs["image"]), _t(arrs["dist"])
    cfg = meta["cases"]["Bounded_L2"]
    tr = MultiscaleTrainer(cfg, image, coords, dist, meta["radii"], tuple(meta["shape"]), dev, seed=meta["seed"])
    got = np.array([s[1] for s in tr.fit(meta["steps"], log_every=1)])
    np.testing.assert_allclose(got, arrs["Bounded_L2/losses"], rtol=5e-4)
    torch.testing.assert_close(tr.predict_all().cpu(), _t(arrs["Bounded_L2/final_out"]), rtol=2e-3, atol=5e-5)


def test_multiscale_trajectory_golden(dev):
    """LogSpace (x0.5) + 0.1*Consistency over sequential batches: train_kspace_multiscale.py:164-195."""
    from inr_mi355x.train_kspace_multiscale import MultiscaleTrainer
    arrs = _load("trajectory_ms.npz")
    meta = json.load(open(os.path.join(GOLD, "trajectory_ms_meta.json")))
    coords, image, dist = _t(arrs["coords"]), _t(arrs["image"]), _t(arrs["dist"])
    cfg = meta["cases"]["MS_LSL"]
    tr = MultiscaleTrainer(cfg, image, coords, dist, meta["radii"], tuple(meta["shape"]), dev, seed=meta["seed"])
    got = np.array([s[1] for s in tr.fit(meta["steps"], log_every=1)])
    np.testing.assert_allclose(got, arrs["MS_LSL/losses"], rtol=5e-4)
    out = tr.predict_all().cpu()
    torch.testing.assert_close(out, _t(arrs["MS_LSL/final_out"]), rtol=2e-3, atol=5e-5)


@pytest.mark.parametrize("shape", ["fourier_4x256", "multiscale_8x512"])
def test_mfn_full_size_vs_oracle(dev, shape):
    """Widths of the shipped configs (remote/config_fourier_kspace.yaml; BASELINE config 4:
    MultiscaleKFourier 8x512, 64-coordinate tiles).  Criterion: as close to a float64 evaluation as the
    reference's own fp32 CPU path (x4, floor 1e-5), see tests/test_gpu_wire.py."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    from inr_mi355x.mfn import FourierNet, MultiscaleKFourier
    multi = shape.startswith("multi")
    net = dict(network_input_size=512, network_output_size=2, network_depth=8 if multi else 4,
               network_width=512 if multi else 256)
    enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
    torch.manual_seed(0)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    model = (MultiscaleKFourier if multi else FourierNet)(net)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).bind_encoder(enc)
    B = 333
    g = torch.Generator().manual_seed(1)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    kind = "MultiscaleKFourier" if multi else "Fourier"
    keys = O.trainable_keys(kind, sd)

    def ref(dtype):
        params = {k: (v.to(dtype).clone().requires_grad_(True) if k in keys else v.to(dtype)) for k, v in sd.items()}
        x = O.encode(coords.to(dtype), enc.B.cpu().to(dtype), "gauss")
        outs = O.model_forward(kind, params, x, net)
        outs = outs if isinstance(outs, list) else [outs]
        loss = sum(O.loss_l2_half(o, gt.to(dtype)) for o in outs)
        grads = torch.autograd.grad(loss, [params[k] for k in keys])
        return torch.stack([o.detach() for o in outs]), loss.detach(), torch.cat([x_.reshape(-1) for x_ in grads])

    o32, l32, g32 = ref(torch.float32)
    o64, l64, g64 = ref(torch.float64)
    eng = model._engine()
    out = eng.forward(coords.to(dev), enc.B.contiguous(), save=False).cpu()
    loss = eng.train_step(coords.to(dev), enc.B.contiguous(), gt.to(dev), M.LossSpec(L.LOSS_L2_HALF))
    flat = eng.grads.cpu()
    live = torch.cat([flat[o:o + n] for (o, n, s, c), lv in zip(model._layout, model._live) if lv])
    from conftest import record_parity
    for name, got, r32, r64 in (("out", out, o32, o64), ("grad", live, g32, g64)):
        e_gpu, e_cpu = rel_l2(got, r64), rel_l2(r32, r64)
        record_parity(f"mfn_full:{shape}", what=name, e_gpu=e_gpu, e_cpu=e_cpu, e_gpu_vs_cpu32=rel_l2(got, r32))
        assert e_gpu <= max(4 * e_cpu, 1e-5), (name, e_gpu, e_cpu)
    assert abs(float(loss) - float(l64)) <= max(4 * abs(float(l32) - float(l64)), 1e-5 * abs(float(l64)))


def test_wide_bounded_gradient_vs_oracle(dev):
    """MultiscaleBoundedFourier at width 512 (two-waves-per-group kernel + batch dW GEMM): BoundedLinear zeroes
    only the Linear's INPUT rows (mfn.py:281-286), so db_i = sum over ALL rows of g_l while dW_i sees the masked
    h_i.  dist values on both sides of every bound; per-tensor comparison so a wrong bias cannot hide in the norm."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    from inr_mi355x.mfn import MultiscaleBoundedFourier
    bounds = META["BoundedFourier"]["bounds8"]
    net = dict(network_input_size=512, network_output_size=2, network_depth=8, network_width=512)
    enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
    torch.manual_seed(0)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    model = MultiscaleBoundedFourier(net, boundaries=bounds)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).bind_encoder(enc)
    B = 300
    g = torch.Generator().manual_seed(1)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
    for lo, hi in bounds:  # every bound splits the batch
        inside = int(((dist >= lo) & (dist <= hi)).sum())
        assert hi >= 5.0 or 0 < inside < B, (lo, hi, inside)
    keys = O.trainable_keys("BoundedFourier", sd)

    def ref(dtype):
        params = {k: (v.to(dtype).clone().requires_grad_(True) if k in keys else v.to(dtype)) for k, v in sd.items()}
        x = O.encode(coords.to(dtype), enc.B.cpu().to(dtype), "gauss")
        outs = O.model_forward("BoundedFourier", params, x, net, dist_to_center=dist.to(dtype), boundaries=bounds)
        loss = sum(O.loss_l2_half(o, gt.to(dtype)) for o in outs)
        grads = torch.autograd.grad(loss, [params[k] for k in keys])
        return loss.detach(), dict(zip(keys, grads))

    l32, g32 = ref(torch.float32)
    l64, g64 = ref(torch.float64)
    eng = model._engine()
    loss = eng.train_step(coords.to(dev), enc.B.contiguous(), gt.to(dev), M.LossSpec(L.LOSS_L2_HALF), dist=dist.to(dev))
    flat = eng.grads.cpu()
    names = [k for k, _ in model.named_parameters()]
    assert abs(float(loss) - float(l64)) <= max(4 * abs(float(l32) - float(l64)), 1e-5 * abs(float(l64)))
    bad = []
    for name, (o, n, s, c), lv in zip(names, model._layout, model._live):
        if not lv:
            continue
        got = flat[o:o + n].view(s)
        e_gpu, e_cpu = rel_l2(got, g64[name]), rel_l2(g32[name], g64[name])
        if e_gpu > max(4 * e_cpu, 1e-5):
            bad.append((name, e_gpu, e_cpu))
    assert not bad, bad


@pytest.mark.parametrize("kind", ["multiscale", "gabor", "bounded"])
def test_wide_mfn_persistent_blocks_additivity(dev, kind):
    """The 512-wide kernel at a batch where every workgroup walks several tiles (40 000 rows = 625 tiles on 256
    workgroups): its slabs carry running sums from tile to tile.  Gradient and loss must equal the sum over three
    pieces that fit one tile per workgroup -- the path test_mfn_full_size_vs_oracle pins against the oracle."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    from inr_mi355x.mfn import GaborNet, MultiscaleKFourier, MultiscaleBoundedFourier
    net = dict(network_input_size=512, network_output_size=2, network_depth=8 if kind != "gabor" else 3,
               network_width=512)
    enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
    torch.manual_seed(0)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    cls = dict(multiscale=MultiscaleKFourier, gabor=GaborNet,
               bounded=lambda n_: MultiscaleBoundedFourier(n_, boundaries=META["BoundedFourier"]["bounds8"]))[kind]
    model = cls(net).to(dev).bind_encoder(enc)
    eng = model._engine()
    B = 40000
    g = torch.Generator().manual_seed(1)
    coords = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
    gt = (torch.randn(B, 2, generator=g) * 0.2).to(dev)
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2).contiguous()
    spec = M.LossSpec(L.LOSS_L2_HALF)
    nt, nb = eng.launch_dims(B)
    assert nt > 2 * nb
    kw = dict(dist=dist) if kind != "gabor" else {}

    def step(lo, hi):
        k = {n: v[lo:hi] for n, v in kw.items()}
        l = float(eng.train_step(coords[lo:hi], enc.B.contiguous(), gt[lo:hi], spec, count=B, **k))
        return l, eng.grads.clone()

    l_all, g_all = step(0, B)
    cuts = [0, 16000, 32000, B]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        assert eng.launch_dims(hi - lo)[0] <= nb
    parts = [step(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
    l_sum, g_sum = sum(p[0] for p in parts), sum(p[1] for p in parts)
    assert abs(l_sum - l_all) <= 5e-6 * abs(l_all)
    assert rel_l2(g_sum, g_all) < 5e-6
    assert torch.equal(step(0, B)[1], g_all)  # run-to-run determinism


@pytest.mark.parametrize("model", ["Fourier", "Gabor", "KGabor"])
def test_single_scale_trainer_mfn_vs_oracle(dev, model):
    """train.py's registry (train.py:63-68) also builds the filter networks for the single-scale loop:
    INRTrainer against the oracle's restatement of that loop, same seeds, masked rows included."""
    from inr_mi355x.synthetic import make_kspace
    from inr_mi355x.train import INRTrainer
    C, H, W = 2, 24, 20
    image, coords, shape = make_kspace(C, H, W)
    cfg = dict(model=model, loss="L2", lr=1e-3, batch_size=300, max_epoch=2, weight_decay=0.0, beta1=0.9, beta2=0.999,
               net=dict(network_input_size=32, network_output_size=2, network_depth=3, network_width=48),
               encoder=dict(embedding="gauss", scale=2, embedding_size=16, coordinates_size=3))
    mask = torch.rand(coords.shape[0], generator=torch.Generator().manual_seed(2)) < 0.7
    tr = INRTrainer(cfg, image, coords, shape, dev, seed=3, mask=mask)
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    ref = O.train_single_scale(cfg, sd, tr.encoder.B.cpu(), coords, image, 7, mask=mask)
    got = np.array([s[1] for s in tr.fit(7, log_every=1)])
    np.testing.assert_allclose(got, np.array(ref), rtol=5e-5)
    out = tr.predict_all().cpu()
    with torch.no_grad():
        want = O.model_forward(model, sd, O.encode(coords, tr.encoder.B.cpu(), "gauss"), cfg["net"])
    torch.testing.assert_close(out, want, rtol=1e-3, atol=2e-5)


def test_multiscale_trainer_computes_its_radii(dev):
    """radii=None: the trainer runs the ring partition itself (train_kspace_multiscale.py:73-84) on the device
    tensors and gets what the host computation gives; then it trains."""
    from inr_mi355x import clustering as Cl
    from inr_mi355x.synthetic import make_kspace
    from inr_mi355x.train_kspace_multiscale import MultiscaleTrainer
    C, H, W = 2, 48, 40
    image, coords, shape = make_kspace(C, H, W)
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2)
    cfg = dict(model="MultiscaleKFourier", loss="LSL", loss_opts=dict(eps=3e-3), lr=3e-4, batch_size=1000, max_epoch=2,
               weight_decay=0.0, beta1=0.9, beta2=0.999, partition=dict(no_steps=40, no_models=4),
               net=dict(network_input_size=32, network_output_size=2, network_depth=8, network_width=32),
               encoder=dict(embedding="gauss", scale=2, embedding_size=16, coordinates_size=3))
    tr = MultiscaleTrainer(cfg, image, coords, dist, None, shape, dev, seed=0)
    _, radii = Cl.partition_kspace(image.reshape(C, H, W, 2), coords.reshape(C, H, W, 3), 40, 4)
    np.testing.assert_allclose(tr.radii, radii, rtol=1e-6)
    assert tr.mx.shape == (5,) and float(tr.mx[-1]) == 1.0
    losses = [s[1] for s in tr.fit(6, log_every=1)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_multiscale_percoil_mask_tv_golden(dev):
    """Per-coil batches + grid undersampling + TV on the last head in the multiscale loop
    (train_kspace_multiscale.py:164-195 with use_tv and a mask): reference-driven trajectory; the trainer builds the
    mask and the zero-filled k-space from the config string and runs the unfused forward / multi-head loss / TV /
    backward step."""
    from inr_mi355x.train_kspace_multiscale import MultiscaleTrainer
    arrs = _load("trajectory_ms.npz")
    meta = json.load(open(os.path.join(GOLD, "trajectory_ms_meta.json")))
    cfg = meta["cases"]["MS_percoil_tv"]
    C, H, W = meta["shape"]
    coords, image, dist = _t(arrs["coords"]), _t(arrs["image"]), _t(arrs["dist"])
    tr = MultiscaleTrainer(cfg, image, coords, dist, meta["radii"], (C, H, W), dev, seed=meta["seed"])
    assert tr.use_tv and tr.bs == H * W and torch.equal(tr.mask_cpu, _t(arrs["mask"]))
    got = np.array([s[1] for s in tr.fit(log_every=1)])
    np.testing.assert_allclose(got, arrs["MS_percoil_tv/losses"], rtol=5e-5)
    torch.testing.assert_close(tr.predict_all().cpu(), _t(arrs["MS_percoil_tv/final_out"]), rtol=1e-3, atol=5e-5)
    # masked but fused (no TV): the fused kernel's mask gates the pointwise terms only, like the tier-1 loss kernel
    cfg2 = dict(cfg, use_tv=False)
    a = MultiscaleTrainer(cfg2, image, coords, dist, meta["radii"], (C, H, W), dev, seed=meta["seed"])
    sd = {k: v.detach().cpu().clone() for k, v in a.model.state_dict().items()}
    want = O.train_multiscale(cfg2, sd, a.encoder.B.cpu(), coords, image * _t(arrs["mask"])[:, None], dist, meta["radii"],
                              4, mask=_t(arrs["mask"]), grid_hw=(H, W))
    got2 = np.array([s[1] for s in a.fit(4, log_every=1)])
    np.testing.assert_allclose(got2, np.array(want), rtol=5e-5)
