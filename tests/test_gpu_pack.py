"""The two ways inr_pack_params / inr_adam_step refresh the MFMA fragment images of plain-layer networks (``-m gpu``):
entry by entry (adam_pack_kernel: one flat parameter -> its image words) and image by image (pack_images_real_kernel: one
float4 image slot -> the parameters it holds; taken for networks of >= 2^18 parameters, forced here by INR_PACK_BY_IMAGE).
No reference counterpart (the images are what ATen's GEMM packing does implicitly); the two must agree bit for bit on every
word of the packed buffer, padding included, and so must the Adam update in front of them (torch.optim.Adam, train.py:76,190)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_pack():
    old = os.environ.get("INR_PACK_BY_IMAGE")

    def set_(v):
        if v is None:
            os.environ.pop("INR_PACK_BY_IMAGE", None)
        else:
            os.environ["INR_PACK_BY_IMAGE"] = v

    yield set_
    set_(old)


def _engines():
    import inr_mi355x as M
    from inr_mi355x.mfn import FourierNet, MultiscaleKFourier
    dev = torch.device("cuda:0")
    torch.manual_seed(3)

    def net(depth, width, in_f, out_f=2):
        return dict(network_input_size=in_f, network_output_size=out_f, network_depth=depth, network_width=width,
                    last_tanh=False)

    def enc(E):
        return M.Positional_Encoder(dict(embedding="gauss", scale=2, embedding_size=E, coordinates_size=3), device=dev)

    # SIREN on encoded input (natural k order everywhere), ragged widths and inputs
    yield "siren_x_5x200_in70", M.SIREN(net(5, 200, 70)).to(dev)._engine()
    yield "ffn_x_3x129_in33_out3", M.FFN(net(3, 129, 33, 3)).to(dev)._engine()
    # fused gauss encoder (layer 0 in split k order), the 512-wide two-wave kernels
    yield "siren_gauss_4x512_E256", M.SIREN(net(4, 512, 512)).to(dev).fused_engine(256)
    yield "siren_gauss_3x160_E48", M.SIREN(net(3, 160, 96)).to(dev).fused_engine(48)
    # filter networks: filters in split k order without a transposed image, linears, four heads (BASELINE config 4's shape)
    m = MultiscaleKFourier(net(8, 512, 512))
    m = m.to(dev)
    m.bind_encoder(enc(256))
    yield "multiscale_gauss_8x512_E256", m._engine("gauss")
    m = FourierNet(net(3, 96, 40)).to(dev)
    yield "fourier_x_3x96_in40", m._engine("x")


def test_pack_by_image_equals_pack_by_entry(force_pack):
    assert torch.cuda.is_available()
    for name, eng in _engines():
        if getattr(eng, "desc", None) is not None and eng.desc.precision != 0:
            continue
        force_pack("0")
        eng.packed.fill_(float("nan"))
        eng.pack()
        by_entry = eng.packed.clone()
        force_pack("1")
        eng.packed.fill_(float("nan"))
        eng.pack()
        by_image = eng.packed.clone()
        # words no parameter maps to are padding: the entry-wise pack leaves them alone (zeros from the allocation; NaN
        # here), the image-wise pack writes zeros
        pad = torch.isnan(by_entry)
        assert torch.equal(by_image[~pad], by_entry[~pad]), name
        assert bool((by_image[pad] == 0).all()), name
        assert int((~pad).sum()) >= eng.n_params, name
        # the update in front: same arithmetic, two launches instead of one
        g = torch.Generator(device="cpu").manual_seed(5)
        grads = (torch.randn(eng.n_params, generator=g) * 1e-3).cuda()
        p0 = eng.params.clone()
        res = {}
        for mode in ("0", "1"):
            force_pack(mode)
            eng.params.copy_(p0)
            eng.exp_avg.zero_()
            eng.exp_avg_sq.zero_()
            eng.step = 0
            eng.packed.zero_()
            eng.grads.copy_(grads)
            for _ in range(3):
                eng.adam_step(1e-3, weight_decay=1e-4)
            res[mode] = (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), eng.packed.clone())
        for a, b in zip(res["0"], res["1"]):
            assert torch.equal(a, b), name
        assert not torch.equal(res["0"][0], p0)
        force_pack(None)
