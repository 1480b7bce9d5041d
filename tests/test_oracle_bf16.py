"""The bf16 path's rounding model (oracle/inr_oracle_bf16.py) on the CPU: its 8-bit formats do what the device measured
(tools/probes/fmt8_probe.hip, bf8_clamp_probe.hip -- outputs under profiles/), and the model as a whole stays within a few
percent of the fp32 oracle (which is pinned to the reference): the rounding model is the reference's arithmetic plus
roundings, nothing else."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle as O  # noqa: E402

NET = dict(network_input_size=64, network_output_size=2, network_depth=4, network_width=256, last_tanh=True)


def test_phase_byte_is_round_to_nearest_even_mod_256():
    t = torch.tensor([0.0, 0.5 / 256, 1.5 / 256, 2.5 / 256, 1.0, 1.0 + 3.0 / 256, -1.0 / 256, -0.5 / 256, -1.5 / 256, 255.998,
                      -255.998, 0.7491])
    want = [0, 0, 2, 2, 0, 3, 255, 0, 254, 255, 1, round(0.7491 * 256) % 256]  # (255.998 * 256 = 65535.49)
    assert O.bf16.phase_byte(t).tolist() == [float(v) for v in want]


def test_bf8_is_e5m2_nearest_even_saturating_with_subnormals():
    x = torch.tensor([1.0, 1.125, 1.375, 1.625, 1.875, -1.3, 57344.0, 60000.0, 1e9, -1e9, 1.5259e-5, 7.6e-6, -7.7e-6])
    want = [1.0, 1.0, 1.5, 1.5, 2.0, -1.25, 57344.0, 57344.0, 57344.0, -57344.0, 2.0 ** -16, 0.0, -(2.0 ** -16)]
    assert O.bf16.bf8(x).tolist() == want


def test_rounding_model_stays_near_the_fp32_oracle():
    torch.manual_seed(0)
    sd = O.init_siren(NET)
    enc_B = torch.randn(32, 3) * 2.0
    g = torch.Generator().manual_seed(1)
    B = 600
    coords = torch.rand(B, 3, generator=g) * 2 - 1
    gt = torch.randn(B, 2, generator=g) * 0.2
    mult = 2.0 ** 12 * B
    y, grads, amax = O.bf16.siren_bf16_step(sd, coords, enc_B, NET, lambda yy: (yy - gt) / (B * 2.0), mult)
    _, grads_w, _ = O.bf16.siren_bf16_step(sd, coords, enc_B, NET, lambda yy: (yy - gt) / (B * 2.0), mult, wide_sums=True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.siren_forward(params, O.encode(coords, enc_B, "gauss"), NET)
    ref = dict(zip(params, torch.autograd.grad(O.loss_l2_half(out, gt), list(params.values()))))
    assert float((y - out.detach()).abs().max()) < 3e-2
    assert 1.0 < amax < 57344.0
    for k in ref:
        e = float((grads[k].double() - ref[k].double()).norm() / ref[k].double().norm())
        e_self = float((grads[k].double() - grads_w[k].double()).norm() / grads[k].double().norm())
        assert e < 1e-1, (k, e)          # the roundings cost a few percent per tensor ...
        assert e_self < 2e-2, (k, e_self)  # ... and leave the summation order a much smaller say
    # the same step under another power-of-two scale: the same gradients up to what the smallest dZ (bf8 subnormals, 2^-16
    # .. 2^-14 of the scaled range) make of it
    _, g2, _ = O.bf16.siren_bf16_step(sd, coords, enc_B, NET, lambda yy: (yy - gt) / (B * 2.0), mult * 4.0)
    for k in ref:
        assert float((g2[k].double() - grads[k].double()).norm() / grads[k].double().norm()) < 1e-3, k
