"""The row-split fused step (csrc/inr_mlp_rs_impl.h) through the C-ABI (``-m gpu``).

inr_train_step of a SIREN / FFN plan behind the fused gauss encoder (hidden width 129..256) runs one of two fused kernels:
the row-split kernel (the four waves of a workgroup split the output rows and share 16-coordinate column blocks) or
inr_mlp_kernel (one wave per 32 coordinates) -- chosen per batch (inr_plan_step_info), forced by INR_RS=1 / INR_RS=0.
Both must produce the reference's loss and gradients (models/networks.py:23-35, 48-69, 91-124; loop train.py:158-192):
rtol 1e-5 on the loss, 1e-5 relative L2 per parameter tensor against the CPU oracle, and the same between the two kernels.
"""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402  (checker only)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture
def force_rs():
    """sets INR_RS for the duration of a test (the library reads it per call)"""
    old = os.environ.get("INR_RS")

    def set_(v):
        if v is None:
            os.environ.pop("INR_RS", None)
        else:
            os.environ["INR_RS"] = v

    yield set_
    set_(old)


def _net(depth=5, width=256, E=256, out_f=2, last_tanh=False):
    return (dict(network_input_size=2 * E, network_output_size=out_f, network_depth=depth, network_width=width,
                 last_tanh=last_tanh),
            dict(embedding="gauss", scale=2, embedding_size=E, coordinates_size=3))


def _oracle(kind, sd, encB, coords, gt, net, mask, loss_name, hdr=None):
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = O.encode(coords, encB, "gauss")
    out = (O.siren_forward if kind == "SIREN" else O.ffn_forward)(params, x, net)
    o, g = (out, gt) if mask is None else (out[mask], gt[mask])
    if loss_name == "L2":
        loss = O.loss_l2_half(o, g)
    else:  # (unmasked only: A of HDRLoss_FF is a mean over all batch coordinates)
        assert mask is None
        loss = O.loss_hdr(o, g, coords, hdr)[0]
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), [x.detach() for x in grads]


def _case(dev, force_rs, B, kind="SIREN", masked=False, loss_name="L2", **net_kw):
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    net, enc_cfg = _net(**net_kw)
    torch.manual_seed(B % 1000 + net["network_depth"])
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    model = (M.SIREN if kind == "SIREN" else M.FFN)(net)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    g = torch.Generator().manual_seed(B)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, net["network_output_size"], generator=g) * 0.2
    mask = torch.rand(B, generator=g) < 0.6 if masked else None
    cnt = B if mask is None else int(mask.sum())
    hdr = dict(hdr_eps=1e-3, hdr_ff_sigma=1.0, hdr_ff_factor=0.5) if loss_name == "HDR" else None
    spec = M.LossSpec.from_config({"loss": loss_name, "loss_opts": hdr})
    # HDRLoss_FF's gradient 2 log(|e| / den) e / |e|^2 amplifies the outputs' rounding where |e| is small: 2e-5 at the golden
    # vectors' batch (tests/test_gpu_parity.py), 4e-5 over 25 000 rows -- for BOTH kernels (errors recorded below)
    tol = 1e-5 if loss_name == "L2" else 4e-5
    eng = model.fused_engine(enc_cfg["embedding_size"])
    m = None if mask is None else mask.to(torch.uint8).to(dev)
    hdr_A = 0.0
    if loss_name == "HDR":
        f = torch.exp(-(coords[:, 1] ** 2 + coords[:, 2] ** 2) / (2 * hdr["hdr_ff_sigma"] ** 2))
        hdr_A = float(torch.mean((1 - f) ** 2))
    res = {}
    for rs in ("0", "1"):
        force_rs(rs)
        info = L.StepInfo()
        L.check(eng.lib.inr_plan_step_info(eng.plan, B, C.byref(info)))
        assert info.row_split == int(rs)
        eng.grads.zero_()
        loss = eng.train_step(coords.to(dev), enc.B.contiguous(), gt.to(dev), spec, count=cnt, mask=m, hdr_A=hdr_A).clone()
        res[rs] = (loss.cpu(), eng.grads.clone().cpu())
        loss2 = eng.train_step(coords.to(dev), enc.B.contiguous(), gt.to(dev), spec, count=cnt, mask=m, hdr_A=hdr_A)
        assert torch.equal(loss2.cpu(), res[rs][0]) and torch.equal(eng.grads.cpu(), res[rs][1]), "not run-to-run identical"
    ref_loss, ref_grads = _oracle(kind, sd, enc.B.cpu(), coords, gt, net, mask, loss_name, hdr)
    from conftest import record_parity
    for rs in ("0", "1"):
        torch.testing.assert_close(res[rs][0], ref_loss, rtol=tol, atol=0)
        record_parity("rs_case", B=B, kind=kind, loss=loss_name, row_split=int(rs),
                      e_grad=rel_l2(res[rs][1], torch.cat([x.reshape(-1) for x in ref_grads])))
        off = 0
        for p, rg in zip(model.parameters(), ref_grads):
            n = p.numel()
            assert rel_l2(res[rs][1][off:off + n], rg) < tol, (rs, off)
            assert rel_l2(res[rs][1][off:off + n], res["0"][1][off:off + n]) < tol
            off += n
    return res


@pytest.mark.parametrize("B", [1, 16, 17, 127, 129, 4133, 25000, 33545])
def test_row_split_vs_oracle_and_tile_kernel(dev, force_rs, B):
    """graded shape (SIREN 5x256 / gauss-512): one block, ragged tails, slots that straddle two workgroups' tiles,
    workgroups without a tile (4133 rows: 132 tiles of 2 blocks on 256 workgroups), one round of 7 / 6 blocks, two rounds"""
    _case(dev, force_rs, B)


@pytest.mark.parametrize("B", [112, 113, 24576, 28672, 28673])
def test_row_split_tile_boundaries(dev, force_rs, B):
    """exactly one 7-block tile and one coordinate more; every workgroup with exactly 6 / exactly 7 column blocks
    (256 x 6 x 16, 256 x 7 x 16 rows); one coordinate beyond a full round of 7-block tiles (a second round begins)"""
    _case(dev, force_rs, B)


def test_row_split_full_rounds_forced(dev, force_rs):
    """65 536 rows fill inr_mlp_kernel's rounds exactly, so the library keeps that kernel; forced, the row-split kernel
    deals 6 + 6 + 4 column blocks per workgroup over three rounds and must agree"""
    _case(dev, force_rs, 65536)


@pytest.mark.parametrize("depth", [2, 3, 4, 8])
def test_row_split_depths(dev, force_rs, depth):
    _case(dev, force_rs, 3000, depth=depth)


@pytest.mark.parametrize("width,E", [(129, 64), (160, 32), (200, 96), (256, 512)])
def test_row_split_widths_and_encoders(dev, force_rs, width, E):
    """padded hidden rows (width < 256) and encoder sizes from one chunk of 32 phases to sixteen"""
    _case(dev, force_rs, 2777, width=width, E=E)


def test_row_split_masked_hdr_ffn_outputs(dev, force_rs):
    _case(dev, force_rs, 4133, masked=True)
    _case(dev, force_rs, 25000, loss_name="HDR")
    _case(dev, force_rs, 3000, kind="FFN")
    _case(dev, force_rs, 3000, out_f=3)            # four computed last-layer rows (MO = 4)
    _case(dev, force_rs, 3000, out_f=1, last_tanh=True)
    _case(dev, force_rs, 5000, kind="FFN", masked=True, width=255, E=160)
    _case(dev, force_rs, 5000, loss_name="HDR", depth=3, width=130, E=512)


def test_step_info_matches_what_runs(dev, force_rs):
    """the default choice: row-split below 97 % fill of the tile kernel's rounds, and the two choices give the same step"""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    net, enc_cfg = _net()
    model = M.SIREN(net).to(dev)
    eng = model.fused_engine(256)
    force_rs(None)
    info = L.StepInfo()
    for B, want in ((25000, 1), (65536, 0), (100000, 1), (32768, 0), (31000, 1)):
        L.check(eng.lib.inr_plan_step_info(eng.plan, B, C.byref(info)))
        assert info.row_split == want, (B, info.row_split)
