"""Per-coil batches, undersampling masks and the total-variation term on the HIP path -- ``-m gpu``.
Reference: tv_loss (metrics/losses.py:326-343) inside the masked branch of train.py:172-177, coil
batches from MRICoilWrapperDataset (data/nerp_datasets.py:397-441)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402  (checker only)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _engine(dev):
    import inr_mi355x as M
    torch.manual_seed(0)
    net = dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32)
    return M.SIREN(net).to(dev)._engine()


def _tv_ref(img):
    x = img.clone().double().requires_grad_(True)
    loss = O.loss_tv(x)
    (g,) = torch.autograd.grad(loss, x)
    return float(loss.detach()), g


@pytest.mark.parametrize("H,W", [(2, 2), (7, 5), (24, 20), (640, 368)])
def test_tv_grad_vs_oracle(dev, H, W):
    """Value and gradient of tv_loss against float64 autograd of the oracle, including ties (sign(0) = 0)."""
    eng = _engine(dev)
    g = torch.Generator().manual_seed(H * 1000 + W)
    img = torch.randn(H, W, 2, generator=g)
    img[H // 2:, : W // 2] = 0.25  # flat patch: exact ties -> zero subgradient, as torch's l1_loss
    loss_ref, grad_ref = _tv_ref(img)
    out = img.reshape(-1, 2).to(dev)
    dout = torch.zeros(H * W, 2, device=dev)
    eng._loss.zero_()
    eng._loss[0] = 3.0  # the call ADDS to the running loss and to dout
    loss = eng.tv_grad(out, dout, H, W, H)
    assert abs(float(loss) - 3.0 - loss_ref) <= 2e-6 * max(loss_ref, 1.0) + 1e-6
    torch.testing.assert_close(dout.cpu().double().reshape(H, W, 2), grad_ref, rtol=1e-5, atol=1e-12)
    eng.tv_grad(out, dout, H, W, H)
    torch.testing.assert_close(dout.cpu().double().reshape(H, W, 2), 2 * grad_ref, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_tv_row_slabs_sum_to_whole(dev, world):
    """The data-parallel split: rank slabs (own rows + one halo row) sum to the single-GPU result."""
    from inr_mi355x.train import shard_rows
    eng = _engine(dev)
    H, W = 37, 16
    g = torch.Generator().manual_seed(world)
    img = torch.randn(H, W, 2, generator=g).to(dev)
    whole = torch.zeros(H * W, 2, device=dev)
    eng._loss.zero_()
    loss_whole = float(eng.tv_grad(img.reshape(-1, 2), whole, H, W, H))
    acc = torch.zeros(H * W, 2, device=dev)
    loss_sum = 0.0
    for r in range(world):
        y0, y1 = shard_rows(0, H, r, world)
        ye = min(y1 + 1, H)
        d = torch.zeros((ye - y0) * W, 2, device=dev)
        eng._loss.zero_()
        loss_sum += float(eng.tv_grad(img[y0:ye].reshape(-1, 2).contiguous(), d, y1 - y0, W, H))
        acc[y0 * W:ye * W] += d
    assert abs(loss_sum - loss_whole) < 1e-6 * loss_whole
    torch.testing.assert_close(acc, whole, rtol=1e-6, atol=1e-10)


def test_tv_rejects_bad_shapes(dev):
    eng = _engine(dev)
    out = torch.zeros(12, 2, device=dev)
    with pytest.raises(RuntimeError, match="inr_tv_grad"):
        eng.tv_grad(out, out.clone(), 1, 4, 8)  # 3 rows given, 1 owned: more than one halo row
    with pytest.raises(RuntimeError, match="inr_tv_grad"):
        eng.tv_grad(out, out.clone(), 12, 1, 12)  # W < 2


def test_percoil_tv_trajectory_golden(dev):
    """INRTrainer with per_coil + 'grid-3*2' + use_tv reproduces the reference-driven trajectory; the mask
    and zero-filled k-space are built by the trainer itself from the config string."""
    from inr_mi355x.train import INRTrainer
    arrs = dict(np.load(os.path.join(GOLD, "undersampling.npz")))
    meta = json.load(open(os.path.join(GOLD, "undersampling_meta.json")))
    cfg = meta["config"]
    C, H, W = meta["shape"]
    coords = torch.from_numpy(arrs["grid_coords"])
    full = torch.from_numpy(arrs["grid_full"]).reshape(-1, 2)
    tr = INRTrainer(cfg, full, coords, (C, H, W), dev, seed=meta["seed"])
    assert tr.use_tv and tr.bs == H * W and tr.steps_per_epoch == C
    torch.testing.assert_close(tr.image.cpu(), torch.from_numpy(arrs["grid_masked"]).reshape(-1, 2), rtol=0, atol=0)
    got = np.array([s[1] for s in tr.fit(log_every=1)])
    np.testing.assert_allclose(got, arrs["percoil_tv/losses"], rtol=5e-5)
    torch.testing.assert_close(tr.predict_all().cpu(), torch.from_numpy(arrs["percoil_tv/final_out"]),
                               rtol=1e-3, atol=5e-5)


def test_undersampled_fused_step_matches_oracle(dev):
    """Radial mask, no TV, ordinary batches: the fused masked step (forward on all rows, loss on sampled
    rows, train.py:172-177) against the oracle's loop."""
    from inr_mi355x.synthetic import create_coords, make_kspace
    from inr_mi355x.train import INRTrainer
    from inr_mi355x.undersampling import Undersampler
    C, H, W = 2, 32, 24
    image, coords, shape = make_kspace(C, H, W)
    cfg = dict(model="SIREN", loss="L2", lr=1e-4, batch_size=500, max_epoch=2, weight_decay=0.0, beta1=0.9, beta2=0.999,
               undersampling="radial-2",
               net=dict(network_input_size=32, network_output_size=2, network_depth=3, network_width=32),
               encoder=dict(embedding="gauss", scale=2, embedding_size=16, coordinates_size=3))
    tr = INRTrainer(cfg, image, coords, shape, dev, seed=1, mask_seed=4)
    masked, _, gm = Undersampler("radial", seed=4).apply(image.reshape(C, H, W, 2), [2])
    assert torch.equal(tr.mask_cpu, gm[:, 0])
    torch.manual_seed(1)
    B = O.encoder_init(cfg["encoder"])
    sd = O.init_model("SIREN", cfg["net"])
    ref = O.train_single_scale(cfg, sd, B, coords, masked.reshape(-1, 2), 6, mask=gm[:, 0])
    got = np.array([s[1] for s in tr.fit(6, log_every=1)])
    np.testing.assert_allclose(got, np.array(ref), rtol=5e-5)


@pytest.mark.parametrize("halo", [0, 1])
def test_loss_tv_grad_one_pass_matches_the_two_calls(dev, halo):
    """inr_loss_tv_grad (one pass, 256 workgroups) == inr_loss_grad on the owned rows' sampled coordinates followed by
    inr_tv_grad: same formulas, other summation order (losses.py:326-343; train.py:172-182).  With a halo row (a
    data-parallel rank's slab): the halo row takes no pointwise loss although its mask bytes are set."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    H, W = 37, 52
    R_own = 20
    R = R_own + halo
    g = torch.Generator().manual_seed(11 + halo)
    out = (torch.randn(R * W, 2, generator=g) * 0.3).to(dev)
    gt = (torch.randn(R * W, 2, generator=g) * 0.3).to(dev)
    mask = (torch.rand(R * W, generator=g) < 0.3).to(torch.uint8).to(dev)
    net = dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32, last_tanh=True)
    eng = M.SIREN(net).to(dev).fused_engine(8)
    spec = M.LossSpec(L.LOSS_L2_HALF)
    m2 = mask.clone()
    m2[R_own * W:] = 0
    count = int(m2.sum())
    l_a, d_a = eng.loss_grad(spec, out, gt, count, mask=m2)
    l_a = float(eng.tv_grad(out, d_a, R_own, W, H))
    d_a = d_a.clone()
    l_b, d_b = eng.loss_tv_grad(spec, out, gt, count, R_own, W, H, mask=mask)
    assert abs(float(l_b) - l_a) <= 1e-6 * abs(l_a), (float(l_b), l_a)
    torch.testing.assert_close(d_b, d_a, rtol=1e-6, atol=1e-9)
    l_c, d_c = eng.loss_tv_grad(spec, out, gt, count, R_own, W, H, mask=mask)
    assert float(l_c) == float(l_b) and torch.equal(d_c, d_b)


@pytest.mark.parametrize("tag", ["Fourier_percoil_tv", "Gabor_percoil_tv"])
def test_percoil_tv_filter_networks_golden(dev, tag):
    """tv_loss is model-agnostic in the single-scale loop (train.py:172-175): the filter networks on per-coil batches with
    a grid mask, reference-driven (tools/make_golden.py: extra_trajectories)."""
    from inr_mi355x.train import INRTrainer
    arrs = dict(np.load(os.path.join(GOLD, "trajectory_extra.npz")))
    meta = json.load(open(os.path.join(GOLD, "trajectory_extra_meta.json")))
    cfg = meta["cases"][tag]
    C, H, W = meta["shape"]
    coords = torch.from_numpy(arrs["coords"]).reshape(-1, 3)
    full = torch.from_numpy(arrs["full"]).reshape(-1, 2)
    tr = INRTrainer(cfg, full, coords, (C, H, W), dev, seed=meta["seed"])
    assert tr.use_tv and tr.is_mfn and tr.bs == H * W
    torch.testing.assert_close(tr.image.cpu(), torch.from_numpy(arrs["masked"]).reshape(-1, 2), rtol=0, atol=0)
    got = np.array([s[1] for s in tr.fit(meta["steps"], log_every=1)])
    np.testing.assert_allclose(got, arrs[tag + "/losses"], rtol=1e-4)
    torch.testing.assert_close(tr.predict_all().cpu(), torch.from_numpy(arrs[tag + "/final_out"]), rtol=1e-3, atol=5e-5)
