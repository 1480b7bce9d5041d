"""One full-width layer / stage per family through the shipped kernels -- ``-m gpu``.

The full-size WIRE / WIRE2D / filter networks are numerically chaotic in fp32 (tests/test_gpu_wire.py: two fp32
evaluations that differ only in summation order agree to ~1e-3 after five Gabor layers), so their end-to-end tests
hold the device to a multiple of the oracle's own fp32-vs-float64 distance.  That criterion cannot tell a kernel that is
exact per layer from one that is slightly off per layer.  Here chaos cannot compound: depth-minimal plans -- ONE
181-complex WIRE hidden layer, ONE 256-complex WIRE2D layer, ONE 512-wide filter stage, ONE 512-wide BoundedLinear --
at the full width of the BASELINE configs (so the same kernel instantiations: the 12- and 16-block two-waves-per-group
kernels, their Gabor epilogues, the (y, z) stash Jacobian rebuild, the wide filter passes and the batch dW GEMM), held
to 1e-5 relative against the fp32 oracle DIRECTLY -- forward, loss and every parameter tensor's gradient.  The float64
distances of both are recorded next to it (gpurun_out/parity_errors.jsonl)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402  (checker only)
from test_gpu_widths import _ref, rel_l2  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _hold(tag, mdl, eng_grads, out, loss, r32, r64, live=None):
    from conftest import record_parity
    e_out = rel_l2(out, r32[0])
    record_parity("layer:" + tag, what="out", e_gpu=rel_l2(out, r64[0]), e_cpu=rel_l2(r32[0], r64[0]), e_gpu_vs_cpu32=e_out)
    assert e_out <= 1e-5, (tag, "out", e_out)
    assert abs(float(loss) - float(r32[1])) <= 1e-5 * abs(float(r32[1])), (tag, float(loss), float(r32[1]))
    flat = eng_grads.cpu()
    if live is not None:
        flat = torch.cat([flat[o:o + n] for (o, n, s, c), lv in zip(mdl._layout, live) if lv])
    e_g = rel_l2(flat, r32[2])
    record_parity("layer:" + tag, what="grad", e_gpu=rel_l2(flat, r64[2]), e_cpu=rel_l2(r32[2], r64[2]), e_gpu_vs_cpu32=e_g)
    assert e_g <= 1e-5, (tag, "grad", e_g)
    # tensor by tensor (a bias vector is a thousandth of the flat gradient's norm)
    off = 0
    lay = [(o, n) for (o, n, s, c), lv in zip(mdl._layout, live or [True] * len(mdl._layout)) if lv]
    for (o, n) in lay:
        a, b = flat[off:off + n], r32[2][off:off + n]
        off += n
        if float(b.norm()) > 0:
            assert rel_l2(a, b) <= 2e-5, (tag, "tensor at", o, rel_l2(a, b))


@pytest.mark.parametrize("kind,width", [("WIRE", 256), ("WIRE2D", 256)])
def test_one_complex_gabor_layer_full_width(dev, kind, width):
    """network_depth 1: first layer (real weights on the 3 coordinates), ONE hidden complex Gabor layer -- 181 x 181
    complex (WIRE, networks.py:199-204,228) or 256 x 256 with its scale_orth Linear (WIRE2D, wire2d.py:49-60) --, the
    complex output Linear; omega_0 30 / scale_0 15 as config 3; B = 300 on the 64-coordinate tiles (ragged last tile)."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    net = dict(network_input_size=3, network_output_size=2, network_depth=1, network_width=width, first_omega_0=30,
               hidden_omega_0=30, scale=15)
    torch.manual_seed(5)
    mdl = getattr(M, kind)(net)
    sd = {k: v.clone() for k, v in mdl.state_dict().items()}
    mdl = mdl.to(dev)
    B = 300
    g = torch.Generator().manual_seed(6)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    r32 = _ref(kind, sd, net, coords, None, gt, torch.float32)
    r64 = _ref(kind, sd, net, coords, None, gt, torch.float64)
    eng = mdl._engine()
    out = eng.forward(coords.to(dev), None, save=False).cpu()[None]
    loss = eng.train_step(coords.to(dev), None, gt.to(dev), M.LossSpec(L.LOSS_L2_HALF))
    _hold(f"{kind}-{width}-depth1", mdl, eng.grads, out, loss, r32, r64)


def test_one_filter_stage_width_512(dev):
    """FourierNet depth 1, width 512, 512 encoder features (config 4's widths): h1 = sin(F1 x + c1) * (L0 sin(F0 x + c0) + d0)
    and the output Linear (mfn.py:34-43,85-94) -- one 512 x 512 Linear between two 512 x 512 filters, the 16-block wide
    kernel and its batch dW GEMM."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    from inr_mi355x.mfn import FourierNet
    net = dict(network_input_size=512, network_output_size=2, network_depth=1, network_width=512)
    enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
    torch.manual_seed(7)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    mdl = FourierNet(net)
    sd = {k: v.clone() for k, v in mdl.state_dict().items()}
    mdl = mdl.to(dev).bind_encoder(enc)
    B = 300
    g = torch.Generator().manual_seed(8)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    r32 = _ref("Fourier", sd, net, coords, enc.B.cpu(), gt, torch.float32)
    r64 = _ref("Fourier", sd, net, coords, enc.B.cpu(), gt, torch.float64)
    eng = mdl._engine()
    out = eng.forward(coords.to(dev), enc.B.contiguous(), save=False).cpu()
    loss = eng.train_step(coords.to(dev), enc.B.contiguous(), gt.to(dev), M.LossSpec(L.LOSS_L2_HALF))
    _hold("Fourier-512-depth1", mdl, eng.grads, out, loss, r32, r64, live=mdl._live)


def test_one_bounded_linear_width_512(dev):
    """MultiscaleBoundedFourier depth 1, width 512: ONE BoundedLinear (mfn.py:281-286: rows of h whose dist lies outside
    [lo, hi] are zeroed before the Linear, the bias still reaches them) between two filters, head 1; dist on both sides of
    the bound."""
    import inr_mi355x as M
    from inr_mi355x import _lib as L
    from inr_mi355x.mfn import MultiscaleBoundedFourier
    net = dict(network_input_size=512, network_output_size=2, network_depth=1, network_width=512)
    enc_cfg = dict(embedding="gauss", scale=4, embedding_size=256, coordinates_size=3)
    bounds = [(0.2, 0.9)]
    torch.manual_seed(9)
    enc = M.Positional_Encoder(enc_cfg, device=dev)
    mdl = MultiscaleBoundedFourier(net, boundaries=bounds)
    sd = {k: v.clone() for k, v in mdl.state_dict().items()}
    mdl = mdl.to(dev).bind_encoder(enc)
    B = 300
    g = torch.Generator().manual_seed(10)
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2)
    assert int((dist < 0.2).sum()) > 5 and int((dist > 0.9).sum()) > 5

    names = [n for n, _ in mdl.named_parameters()]
    assert len(names) == len(mdl._live)

    def ref(dtype):  # (O.trainable_keys knows the depth-8 multiscale shapes only: here the live set is the model's own)
        params = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        x = O.encode(coords.to(dtype), enc.B.cpu().to(dtype), "gauss")
        outs = O.bounded_forward(params, x, net, dist.to(dtype), bounds)
        loss = sum(O.loss_l2_half(o.contiguous(), gt.to(dtype)) for o in outs)
        live_names = [n for n, lv in zip(names, mdl._live) if lv]
        gr = torch.autograd.grad(loss, [params[n] for n in live_names])
        dead = torch.autograd.grad(loss, [params[n] for n, lv in zip(names, mdl._live) if not lv], allow_unused=True)
        assert all(g_ is None for g_ in dead)  # what the engine skips really has no gradient in the reference's graph
        return torch.stack([o.detach() for o in outs]), loss.detach(), torch.cat([g_.reshape(-1) for g_ in gr])

    r32, r64 = ref(torch.float32), ref(torch.float64)
    eng = mdl._engine()
    out = eng.forward(coords.to(dev), enc.B.contiguous(), save=False, dist=dist.to(dev)).cpu()
    loss = eng.train_step(coords.to(dev), enc.B.contiguous(), gt.to(dev), M.LossSpec(L.LOSS_L2_HALF), dist=dist.to(dev))
    _hold("BoundedFourier-512-depth1", mdl, eng.grads, out, loss, r32, r64, live=mdl._live)
