"""oracle/inr_oracle_bf16.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py for who may import this package).

CPU restatement of the SIREN step *as the bf16 throughput path computes it*: the reference's arithmetic
(SirenLayer.forward, networks.py:91-96; its adjoint) with a rounding at every place where the
MI355X kernels round (csrc/inr_siren_bf16_impl.h, csrc/inr_dw_gemm_bf16.hip, csrc/inr_w2.h) and nowhere else:

  forward   operands of every GEMM in bf16: bf16(W_l * w0 / 2 pi) -- the sine layers' weights carry the factor, their
            accumulators are phases in revolutions --, bf16 of the encoder features / of h_l = sin(2 pi t_l); fp32
            accumulate; bias b_l * w0 / 2 pi in fp32; last layer bf16(W) and no factor
  stash     P_l = round-to-nearest-even(256 t_l) mod 256 (8 bits of phase);  G_l = bf8 e5m2 of dZ_l * mult
  backward  dZ_last * mult in bf16 against bf16(W^T * w0); dZ_l = dH_l * cos(2 pi P_l / 256) (fp32); the next GEMM's
            operand is bf16(dZ_l)
  dW, db    sum over coordinates of fp16(G_l) x fp16(sin(2 pi P_{l-1} / 256))  (layer 0: fp16 of the fp32 encoder
            features, regenerated with one phase chain per frequency; last layer: fp16(dZ_last * mult)), fp32
            accumulate, divided by mult

What stays different from the device: the order of the fp32 sums and the last bit of sin / cos (hardware v_sin_f32 against
libm).  A value that lands within that noise of a rounding boundary rounds the other way (a 1-ulp difference of one bf16 /
bf8 / phase value); the tests' tolerances are sized for that and nothing more.  Not pinned to the reference -- the
reference has no reduced-precision path; this file pins the KERNELS to a written-down rounding model, and the fp32 oracle
(pinned) bounds the model's distance from the reference.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor

SIREN_W0 = 30.0
INV_2PI_F32 = np.float32(0.15915494309189535)


def _bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def _f16(x: Tensor) -> Tensor:
    return x.to(torch.float16).to(torch.float32)


def bf8(x: Tensor) -> Tensor:
    """fp32 -> e5m2 (round to nearest even, subnormals kept; finite values beyond the largest finite one, 57344, saturate
    there: v_cvt_pk_bf8_f32 under MODE.FP16_OVFL as measured by tools/probes/fmt8_probe.hip / bf8_clamp_probe.hip) -> fp32."""
    return torch.clamp(x, -57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float32)


def phase_byte(t: Tensor) -> Tensor:
    """round-to-nearest-even(256 t) mod 256 the way the kernel gets it: the low 8 bits of fp32(t + 1.5 * 2^15)."""
    s = (t.to(torch.float32) + torch.tensor(49152.0, dtype=torch.float32)).contiguous()
    return (s.view(torch.int32) & 255).to(torch.float32)


def _rev_sin(rev: Tensor) -> Tensor:
    return torch.sin((2.0 * np.pi) * rev.double()).float()


def _rev_cos(rev: Tensor) -> Tensor:
    return torch.cos((2.0 * np.pi) * rev.double()).float()


def gauss_features_gemm(coords: Tensor, enc_B: Tensor) -> Tensor:
    """[B, 2E] encoder features (Positional_Encoder.embedding 'gauss', networks.py:30-33) as both bf16 kernels form them
    (csrc/inr_siren_bf16_impl.h layer 0; csrc/inr_dw_gemm_bf16.hip, first-layer units): one fp32 fma chain x0 -> x1 -> x2
    per frequency, t = frac(x . B_j) in revolutions, then sin(2 pi t) and cos(2 pi t)."""
    x, Bm = coords.double(), enc_B.double()
    t = torch.zeros((coords.shape[0], Bm.shape[0]), dtype=torch.float64)
    for k in range(3):
        t = (x[:, k:k + 1] * Bm[None, :, k] + t).float().double()
    t = t.float()
    t = t - torch.floor(t)
    return torch.cat([_rev_sin(t), _rev_cos(t)], dim=1)


def _mm(a: Tensor, b: Tensor, wide: bool) -> Tensor:
    """fp32 GEMM with fp32 accumulation (the device's) or, ``wide``, with float64 accumulation: the same roundings at
    every stated place, another order of the sums in between."""
    return (a.double() @ b.double()).float() if wide else a @ b


def siren_bf16_step(sd: Dict[str, Tensor], coords: Tensor, enc_B: Tensor, net: dict, dloss_dy, mult: float,
                    mask: Optional[Tensor] = None, wide_sums: bool = False):
    """One gradient step of the bf16 path.  ``dloss_dy(y) -> g`` [B,out] is d(loss)/d(out) (rows outside ``mask`` are
    zeroed here); ``mult`` the factor the kernel multiplied it by (engine.grad_scale_state()[2] after the step).
    Returns (out [B,out], grads dict in state_dict keys, largest |dZ * mult| seen).

    ``wide_sums``: accumulate the forward / backward GEMMs in float64.  The rounding model does not say in which order
    the fp32 sums run; two evaluations that differ only there disagree by what the coarse roundings (a bf8 value that
    lands on the other side of a boundary moves by 12-25 %) make of 1e-7 -- 1e-4 for the last layer's gradient, 4e-3 for the
    first layer's weights, whatever the batch size.  The tests hold the device to a small multiple of THAT distance."""
    D = net["network_depth"]
    last_tanh = net.get("last_tanh", False)
    assert net.get("network_last_linear", True) or last_tanh
    kr = np.float32(SIREN_W0) * INV_2PI_F32  # w0 / 2 pi as the packing kernel forms it (fp32 product)
    W = [sd[f"model.{k}.linear.weight"].float() for k in range(D)]
    b = [sd[f"model.{k}.linear.bias"].float() for k in range(D)]
    feat32 = gauss_features_gemm(coords, enc_B)  # (since round 4 the forward pass, too, takes sine and cosine from ONE chain)
    h = _bf16(feat32)
    P = []
    for l in range(D - 1):
        A = _bf16(W[l] * float(kr))
        t = _mm(h, A.t(), wide_sums) + b[l] * float(kr)
        P.append(phase_byte(t))
        h = _bf16(_rev_sin(t))
    z = _mm(h, _bf16(W[D - 1]).t(), wide_sums) + b[D - 1]
    y = torch.tanh(z) if last_tanh else z
    dy = (1.0 - y * y) if last_tanh else torch.ones_like(y)
    g = dloss_dy(y.detach())
    if mask is not None:
        g = g * mask.to(g.dtype)[:, None]
    dzl = (g * dy * mult).float()
    grads = {}
    amax = 0.0
    hs = [_f16(_rev_sin(p / 256.0)) for p in P]  # the GEMM's B operands: fp16 sine of the stashed phase
    f16_feat = _f16(gauss_features_gemm(coords, enc_B))
    a_last = _f16(dzl)
    grads[f"model.{D - 1}.linear.weight"] = (a_last.double().t() @ hs[D - 2].double() / mult).float()
    grads[f"model.{D - 1}.linear.bias"] = (a_last.double().sum(0) / mult).float()
    dH = _mm(_bf16(dzl), _bf16(W[D - 1] * SIREN_W0), wide_sums)
    for l in range(D - 2, -1, -1):
        dZ = dH * _rev_cos(P[l] / 256.0)
        amax = max(amax, float(dZ.abs().max()))
        G = bf8(dZ)
        left = hs[l - 1] if l > 0 else f16_feat
        grads[f"model.{l}.linear.weight"] = (G.double().t() @ left.double() / mult).float()
        grads[f"model.{l}.linear.bias"] = (G.double().sum(0) / mult).float()
        if l > 0:
            dH = _mm(_bf16(dZ), _bf16(W[l] * SIREN_W0), wide_sums)
    return y, grads, amax
