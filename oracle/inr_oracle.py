"""CPU oracle for the INR fitting hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Plain PyTorch-CPU fp32 restatement of the reference's per-step path
(encoder -> model forward -> loss -> backward -> Adam) written from the
formulas in SURVEY.md Appendix A.  Models are *functional*: a model is an
``OrderedDict`` with the reference's ``state_dict`` keys (Appendix B) plus a
forward function; initialisation replays the reference constructors' RNG
consumption order so that ``torch.manual_seed(s)`` gives bit-identical initial
weights (verified by tests/test_oracle_golden.py against SHA-256 fixtures).

All ``file:line`` citations are into /root/reference/src.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
TWO_PI = 2.0 * np.pi

__all__ = [
    "encoder_init", "encode", "create_coords",
    "init_siren", "siren_forward", "siren_forward_backward_manual",
    "init_ffn", "ffn_forward",
    "init_wire", "wire_forward", "init_wire2d", "wire2d_forward",
    "init_fourier", "fourier_forward", "init_gabor", "gabor_forward", "kgabor_forward",
    "init_multiscale", "multiscale_forward", "init_bounded", "bounded_forward",
    "init_model", "model_forward", "trainable_keys",
    "loss_l2_half", "loss_l1_half", "loss_hdr", "loss_center", "loss_tanh", "loss_logspace", "loss_msle",
    "loss_consistency", "loss_tv", "reg_l1", "reg_l2", "make_loss",
    "adam_init", "adam_step", "lr_factor",
    "complex_abs", "rss", "fft2c", "ifft2c", "psnr",
    "train_single_scale", "train_multiscale", "create_pairs", "reconstruct", "PerItemDataset",
]


# --------------------------------------------------------------------------------------
# Encoder  (models/networks.py:7-35)
# --------------------------------------------------------------------------------------
def encoder_init(params: dict) -> Optional[Tensor]:
    """Positional_Encoder.__init__ (models/networks.py:8-21). Consumes the global torch RNG
    exactly like the reference: one ``torch.randn((E, C))`` for 'gauss', nothing otherwise."""
    kind = params["embedding"]
    if kind == "gauss":
        return torch.randn((params["embedding_size"], params["coordinates_size"])) * params["scale"]
    if kind == "LogF":
        steps = int(params["embedding_size"] / (2 * params["coordinates_size"]))
        return (2.0 ** torch.linspace(0.0, params["scale"], steps=steps)).reshape(-1, 1)
    if kind == "none":
        return None
    raise NotImplementedError(kind)


def encode(x: Tensor, B: Optional[Tensor], kind: str) -> Tensor:
    """Positional_Encoder.embedding (models/networks.py:23-35)."""
    if kind == "LogF":
        parts = []
        for a in range(3):
            p = (TWO_PI * x[:, a:a + 1]) @ B.T
            parts.append(torch.cat((torch.sin(p), torch.cos(p)), dim=-1))
        return torch.cat(parts, dim=-1)
    if B is not None:
        p = (TWO_PI * x) @ B.t()
        return torch.cat([torch.sin(p), torch.cos(p)], dim=-1)
    return x


def create_coords(c: int, h: int, w: int) -> Tensor:
    """data/utils.py:98-108 -- (coil, y, x) meshgrid in [-1,1]^3, C-major flattened."""
    Z, Y, X = torch.meshgrid(torch.linspace(-1, 1, c), torch.linspace(-1, 1, h),
                             torch.linspace(-1, 1, w), indexing="ij")
    return torch.hstack((Z.reshape(-1, 1), Y.reshape(-1, 1), X.reshape(-1, 1)))


# --------------------------------------------------------------------------------------
# Initialisation helpers.  nn.Linear's default init is third-party (torch) arithmetic; we
# instantiate torch.nn.Linear itself so that RNG consumption is identical by construction.
# --------------------------------------------------------------------------------------
def _linear(in_f: int, out_f: int, dtype=torch.float) -> Tuple[Tensor, Tensor]:
    lin = torch.nn.Linear(in_f, out_f, dtype=dtype)
    return lin.weight.detach().clone(), lin.bias.detach().clone()


# --------------------------------------------------------------------------------------
# SIREN  (models/networks.py:74-124)
# --------------------------------------------------------------------------------------
SIREN_W0 = 30.0  # SirenLayer default w0=30 for every layer incl. the first (networks.py:75,114-117)


def init_siren(net: dict) -> "OrderedDict[str, Tensor]":
    """SIREN.__init__ + SirenLayer.init_weights (networks.py:85-89,100-119)."""
    D, W = net["network_depth"], net["network_width"]
    dims = [net["network_input_size"]] + [W] * (D - 1) + [net["network_output_size"]]
    sd = OrderedDict()
    for k in range(D):
        in_f, out_f = dims[k], dims[k + 1]
        w, b = _linear(in_f, out_f)
        bound = 1.0 / in_f if k == 0 else np.sqrt(6.0 / in_f) / SIREN_W0
        w.uniform_(-bound, bound)
        sd[f"model.{k}.linear.weight"] = w
        sd[f"model.{k}.linear.bias"] = b
    return sd


def _siren_flags(net: dict) -> Tuple[bool, bool]:
    last_linear = net.get("network_last_linear", True)
    last_tanh = net.get("last_tanh", False)
    return last_linear, last_tanh


def siren_forward(sd: Dict[str, Tensor], x: Tensor, net: dict) -> Tensor:
    """SirenLayer.forward chained (networks.py:91-96,121-124)."""
    D = net["network_depth"]
    last_linear, last_tanh = _siren_flags(net)
    h = x
    for k in range(D):
        z = h @ sd[f"model.{k}.linear.weight"].t() + sd[f"model.{k}.linear.bias"]
        if k == D - 1:
            if last_tanh:
                h = torch.tanh(z)
            elif last_linear:
                h = z
            else:
                h = torch.sin(SIREN_W0 * z)
        else:
            h = torch.sin(SIREN_W0 * z)
    return h


def siren_forward_backward_manual(sd: Dict[str, Tensor], x: Tensor, net: dict, g_out: Tensor):
    """Hand-derived adjoint of SIREN -- the kernel specification.
    Returns (out, grads dict).  ``g_out`` = dL/d(out) [B,out]."""
    D = net["network_depth"]
    last_linear, last_tanh = _siren_flags(net)
    hs, zs = [x], []
    h = x
    for k in range(D):
        z = h @ sd[f"model.{k}.linear.weight"].t() + sd[f"model.{k}.linear.bias"]
        zs.append(z)
        if k == D - 1:
            h = torch.tanh(z) if last_tanh else (z if last_linear else torch.sin(SIREN_W0 * z))
        else:
            h = torch.sin(SIREN_W0 * z)
        hs.append(h)
    out = h
    grads = {}
    g_h = g_out
    for k in reversed(range(D)):
        z = zs[k]
        if k == D - 1:
            if last_tanh:
                g_z = g_h * (1.0 - hs[k + 1] ** 2)
            elif last_linear:
                g_z = g_h
            else:
                g_z = g_h * SIREN_W0 * torch.cos(SIREN_W0 * z)
        else:
            g_z = g_h * SIREN_W0 * torch.cos(SIREN_W0 * z)
        grads[f"model.{k}.linear.weight"] = g_z.t() @ hs[k]
        grads[f"model.{k}.linear.bias"] = g_z.sum(0)
        g_h = g_z @ sd[f"model.{k}.linear.weight"]
    return out, grads


# --------------------------------------------------------------------------------------
# FFN  (models/networks.py:48-69)
# --------------------------------------------------------------------------------------
def init_ffn(net: dict) -> "OrderedDict[str, Tensor]":
    D, W = net["network_depth"], net["network_width"]
    dims = [net["network_input_size"]] + [W] * (D - 1) + [net["network_output_size"]]
    sd = OrderedDict()
    for k in range(D):
        w, b = _linear(dims[k], dims[k + 1])
        sd[f"model.{2 * k}.weight"] = w  # Sequential indices skip the activations
        sd[f"model.{2 * k}.bias"] = b
    return sd


def ffn_forward(sd, x: Tensor, net: dict) -> Tensor:
    D = net["network_depth"]
    h = x
    for k in range(D):
        z = h @ sd[f"model.{2 * k}.weight"].t() + sd[f"model.{2 * k}.bias"]
        h = torch.sigmoid(z) if k == D - 1 else torch.relu(z)
    return h


# --------------------------------------------------------------------------------------
# WIRE / WIRE2D  (models/networks.py:160-260, models/wire2d.py:4-117)
# --------------------------------------------------------------------------------------
def _wire_hidden(net: dict, two_d: bool) -> int:
    W = net["network_width"]
    return W if two_d else int(W / np.sqrt(2))  # networks.py:228 ; wire2d.py does not reduce


def _init_wire_like(net: dict, two_d: bool) -> "OrderedDict[str, Tensor]":
    depth = net["network_depth"]  # number of *hidden* complex layers (networks.py:241-245)
    hid = _wire_hidden(net, two_d)
    sd = OrderedDict()
    for k in range(depth + 1):
        first = k == 0
        in_f = net["network_input_size"] if first else hid
        dtype = torch.float if first else torch.cfloat
        omega = net["first_omega_0"] if first else net["hidden_omega_0"]
        sd[f"net.{k}.omega_0"] = omega * torch.ones(1)
        sd[f"net.{k}.scale_0"] = net["scale"] * torch.ones(1)
        w, b = _linear(in_f, hid, dtype)
        sd[f"net.{k}.linear.weight"], sd[f"net.{k}.linear.bias"] = w, b
        if two_d:
            w2, b2 = _linear(in_f, hid, dtype)
            sd[f"net.{k}.scale_orth.weight"], sd[f"net.{k}.scale_orth.bias"] = w2, b2
    w, b = _linear(hid, net["network_output_size"], torch.cfloat)
    sd[f"net.{depth + 1}.weight"], sd[f"net.{depth + 1}.bias"] = w, b
    return sd


def init_wire(net: dict):
    return _init_wire_like(net, two_d=False)


def init_wire2d(net: dict):
    return _init_wire_like(net, two_d=True)


def wire_forward(sd, x: Tensor, net: dict) -> Tensor:
    """ComplexGaborLayer.forward chained, final complex Linear, ``.real`` (networks.py:199-204,254-258)."""
    depth = net["network_depth"]
    h = x
    for k in range(depth + 1):
        lin = h @ sd[f"net.{k}.linear.weight"].t() + sd[f"net.{k}.linear.bias"]
        omega = sd[f"net.{k}.omega_0"] * lin
        scale = sd[f"net.{k}.scale_0"] * lin
        h = torch.exp(1j * omega - scale.abs().square())
    out = h @ sd[f"net.{depth + 1}.weight"].t() + sd[f"net.{depth + 1}.bias"]
    return out.real


def wire2d_forward(sd, x: Tensor, net: dict) -> Tensor:
    """ComplexGaborLayer2D.forward chained (wire2d.py:49-60,112-117)."""
    depth = net["network_depth"]
    h = x
    for k in range(depth + 1):
        lin = h @ sd[f"net.{k}.linear.weight"].t() + sd[f"net.{k}.linear.bias"]
        orth = h @ sd[f"net.{k}.scale_orth.weight"].t() + sd[f"net.{k}.scale_orth.bias"]
        freq = torch.exp(1j * sd[f"net.{k}.omega_0"] * lin)
        arg = lin.abs().square() + orth.abs().square()
        s0 = sd[f"net.{k}.scale_0"]
        h = freq * torch.exp(-s0 * s0 * arg)
    out = h @ sd[f"net.{depth + 1}.weight"].t() + sd[f"net.{depth + 1}.bias"]
    if net.get("last_tanh", False):
        out = torch.tanh(out)  # complex tanh module appended to the Sequential (wire2d.py:106-107)
    return out.real


# --------------------------------------------------------------------------------------
# Multiplicative filter networks  (models/mfn.py)
# --------------------------------------------------------------------------------------
def _mfn_base_init(sd, hidden: int, out: int, n_layers: int, weight_scale: float):
    """MFNBase.__init__ (mfn.py:15-32): n hidden Linear, output Linear, then weight.uniform_ on each hidden."""
    ws = []
    for i in range(n_layers):
        w, b = _linear(hidden, hidden)
        sd[f"linear.{i}.weight"], sd[f"linear.{i}.bias"] = w, b
        ws.append(w)
    w, b = _linear(hidden, out)
    sd["output_linear.weight"], sd["output_linear.bias"] = w, b
    bound = np.sqrt(weight_scale / hidden)
    for w in ws:
        w.uniform_(-bound, bound)


def _fourier_layer_init(in_f: int, out_f: int, weight_scale: float):
    """FourierLayer.__init__ (mfn.py:50-55)."""
    w, b = _linear(in_f, out_f)
    w *= weight_scale
    b.uniform_(-np.pi, np.pi)
    return w, b


def init_fourier(net: dict, input_scale: float = 2.0, weight_scale: float = 1.0):
    """FourierNet.__init__ (mfn.py:61-83)."""
    n, W = net["network_depth"], net["network_width"]
    sd = OrderedDict()
    _mfn_base_init(sd, W, net["network_output_size"], n, weight_scale)
    for i in range(n + 1):
        w, b = _fourier_layer_init(net["network_input_size"], W, input_scale / np.sqrt(n + 1))
        sd[f"filters.{i}.linear.weight"], sd[f"filters.{i}.linear.bias"] = w, b
    return sd


def fourier_forward(sd, x: Tensor, net: dict) -> Tensor:
    """FourierNet.forward (mfn.py:85-94): h0 = sin(F0 x + c0); h_i = sin(F_i x + c_i) * (L_{i-1} h + d)."""
    n = net["network_depth"]

    def filt(i):
        return torch.sin(x @ sd[f"filters.{i}.linear.weight"].t() + sd[f"filters.{i}.linear.bias"])

    out = filt(0)
    for i in range(1, n + 1):
        out = filt(i) * (out @ sd[f"linear.{i - 1}.weight"].t() + sd[f"linear.{i - 1}.bias"])
    return out @ sd["output_linear.weight"].t() + sd["output_linear.bias"]


def init_gabor(net: dict, input_scale: float = 2.0, weight_scale: float = 1.0,
               alpha: float = 6.0, beta: float = 1.0):
    """GaborNet/KGaborNet.__init__ + GaborLayer.__init__ (mfn.py:100-113,133-162)."""
    n, W, in_f = net["network_depth"], net["network_width"], net["network_input_size"]
    sd = OrderedDict()
    _mfn_base_init(sd, W, net["network_output_size"], n, weight_scale)
    for i in range(n + 1):
        w, b = _linear(in_f, W)
        mu = 2 * torch.rand(W, in_f) - 1
        gamma = torch.distributions.gamma.Gamma(alpha / (n + 1), beta).sample((W,))
        w *= (input_scale / np.sqrt(n + 1)) * torch.sqrt(gamma[:, None])
        b.uniform_(-np.pi, np.pi)
        # parameters of the module itself precede sub-module parameters in state_dict order
        sd[f"filters.{i}.mu"], sd[f"filters.{i}.gamma"] = mu, gamma
        sd[f"filters.{i}.linear.weight"], sd[f"filters.{i}.linear.bias"] = w, b
    return sd


def _gabor_filter(sd, i: int, x: Tensor) -> Tensor:
    """GaborLayer.forward, with_dist_filtering=False (mfn.py:116-131)."""
    mu, gamma = sd[f"filters.{i}.mu"], sd[f"filters.{i}.gamma"]
    D = (x ** 2).sum(-1)[..., None] + (mu ** 2).sum(-1)[None, :] - 2 * x @ mu.T
    lin = x @ sd[f"filters.{i}.linear.weight"].t() + sd[f"filters.{i}.linear.bias"]
    return torch.sin(lin) * torch.exp(-0.5 * D * gamma[None, :])


def gabor_forward(sd, x: Tensor, net: dict) -> Tensor:
    """MFNBase.forward with Gabor filters (mfn.py:34-43)."""
    n = net["network_depth"]
    out = _gabor_filter(sd, 0, x)
    for i in range(1, n + 1):
        out = _gabor_filter(sd, i, x) * (out @ sd[f"linear.{i - 1}.weight"].t() + sd[f"linear.{i - 1}.bias"])
    return out @ sd["output_linear.weight"].t() + sd["output_linear.bias"]


def kgabor_forward(sd, x: Tensor, net: dict, dist_to_center=None) -> Tensor:
    """KGaborNet.forward (mfn.py:195-204): dist_to_center is passed but unused by the filters
    (with_dist_filtering is never enabled), so the arithmetic equals GaborNet's."""
    return gabor_forward(sd, x, net)


MS_OUTPUT_LAYERS = (1, 3, 5, 7)  # mfn.py:223


def init_multiscale(net: dict, weight_scale: float = 1.0):
    """MultiscaleKFourier.__init__ (mfn.py:216-253)."""
    n, W, out = net["network_depth"], net["network_width"], net["network_output_size"]
    sd = OrderedDict()
    _mfn_base_init(sd, W, out, n, weight_scale)
    del sd["output_linear.weight"], sd["output_linear.bias"]  # replaced below; RNG already consumed
    filt = OrderedDict()
    for i in range(n + 1):
        w, b = _fourier_layer_init(net["network_input_size"], W, weight_scale / np.sqrt(n + 1))
        filt[f"filters.{i}.linear.weight"], filt[f"filters.{i}.linear.bias"] = w, b
    for i in range(n + 1):
        w, b = _linear(W, out)
        sd[f"output_linear.{i}.weight"], sd[f"output_linear.{i}.bias"] = w, b
    sd.update(filt)  # 'output_linear' keeps its slot before 'filters' (Appendix B)
    return sd


def multiscale_forward(sd, x: Tensor, net: dict, dist_to_center=None,
                       output_layers: Sequence[int] = MS_OUTPUT_LAYERS) -> List[Tensor]:
    """MultiscaleKFourier.forward (mfn.py:255-267). The reference also evaluates the unused
    last stage (dead compute); it does not affect any output so it is skipped."""
    n = net["network_depth"]

    def filt(i):
        return torch.sin(x @ sd[f"filters.{i}.linear.weight"].t() + sd[f"filters.{i}.linear.bias"])

    outs = []
    out = filt(0)
    for i in range(1, n + 1):
        if i > max(output_layers):
            break
        out = filt(i) * (out @ sd[f"linear.{i - 1}.weight"].t() + sd[f"linear.{i - 1}.bias"])
        if i in output_layers:
            outs.append(out @ sd[f"output_linear.{i}.weight"].t() + sd[f"output_linear.{i}.bias"])
    return outs


def init_bounded(net: dict, weight_scale: float = 1.0):
    """MultiscaleBoundedFourier.__init__ (mfn.py:300-342): base init, then a *fresh* list of
    BoundedLinear (default nn.Linear init, not re-uniformed), filters, output heads."""
    n, W, out = net["network_depth"], net["network_width"], net["network_output_size"]
    tmp = OrderedDict()
    _mfn_base_init(tmp, W, out, n, weight_scale)  # consumed RNG, then discarded by the reference
    sd = OrderedDict()
    for i in range(n):
        w, b = _linear(W, W)
        sd[f"linear.{i}.linear.weight"], sd[f"linear.{i}.linear.bias"] = w, b
    filt = OrderedDict()
    for i in range(n + 1):
        w, b = _fourier_layer_init(net["network_input_size"], W, weight_scale / np.sqrt(n + 1))
        filt[f"filters.{i}.linear.weight"], filt[f"filters.{i}.linear.bias"] = w, b
    for i in range(n + 1):
        w, b = _linear(W, out)
        sd[f"output_linear.{i}.weight"], sd[f"output_linear.{i}.bias"] = w, b
    sd.update(filt)
    return sd


def bounded_forward(sd, x: Tensor, net: dict, dist_to_center: Tensor, boundaries,
                    output_layers: Sequence[int] = MS_OUTPUT_LAYERS) -> List[Tensor]:
    """MultiscaleBoundedFourier.forward + BoundedLinear.forward (mfn.py:281-286,344-355)."""
    n = net["network_depth"]
    dist = dist_to_center.reshape(-1)

    def filt(i):
        return torch.sin(x @ sd[f"filters.{i}.linear.weight"].t() + sd[f"filters.{i}.linear.bias"])

    outs = []
    out = filt(0)
    for i in range(1, n + 1):
        if i > max(output_layers):
            break
        lo, hi = boundaries[i - 1]
        keep = ~((dist < lo) | (dist > hi))
        hb = out * keep[:, None].to(out.dtype)
        out = filt(i) * (hb @ sd[f"linear.{i - 1}.linear.weight"].t() + sd[f"linear.{i - 1}.linear.bias"])
        if i in output_layers:
            outs.append(out @ sd[f"output_linear.{i}.weight"].t() + sd[f"output_linear.{i}.bias"])
    return outs


# --------------------------------------------------------------------------------------
# Model registry (train.py:55-70; train_kspace_multiscale.py:93-101)
# --------------------------------------------------------------------------------------
_INIT = {"SIREN": init_siren, "FFN": init_ffn, "WIRE": init_wire, "WIRE2D": init_wire2d,
         "Fourier": init_fourier, "Gabor": init_gabor, "KGabor": init_gabor,
         "MultiscaleKFourier": init_multiscale, "BoundedFourier": init_bounded}


def init_model(model: str, net: dict):
    return _INIT[model](net)


def model_forward(model: str, sd, x: Tensor, net: dict, dist_to_center=None, boundaries=None):
    if model == "SIREN":
        return siren_forward(sd, x, net)
    if model == "FFN":
        return ffn_forward(sd, x, net)
    if model == "WIRE":
        return wire_forward(sd, x, net)
    if model == "WIRE2D":
        return wire2d_forward(sd, x, net)
    if model == "Fourier":
        return fourier_forward(sd, x, net)
    if model in ("Gabor", "KGabor"):
        return gabor_forward(sd, x, net)
    if model == "MultiscaleKFourier":
        return multiscale_forward(sd, x, net)
    if model == "BoundedFourier":
        return bounded_forward(sd, x, net, dist_to_center, boundaries)
    raise NotImplementedError(model)


def trainable_keys(model: str, sd) -> List[str]:
    """Keys that receive gradients.  omega_0/scale_0 are frozen Parameters (networks.py:191-192);
    MultiscaleKFourier's dead layers never get a grad and Adam skips them."""
    keys = [k for k in sd if not (k.endswith("omega_0") or k.endswith("scale_0"))]
    if model in ("MultiscaleKFourier", "BoundedFourier"):
        n_f = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("filters."))
        live_max = max(MS_OUTPUT_LAYERS)
        out = []
        for k in keys:
            parts = k.split(".")
            idx = int(parts[1])
            if parts[0] == "filters" and idx > live_max:
                continue
            if parts[0] == "linear" and idx > live_max - 1:
                continue
            if parts[0] == "output_linear" and idx not in MS_OUTPUT_LAYERS:
                continue
            out.append(k)
        assert n_f >= live_max + 1
        return out
    return keys


# --------------------------------------------------------------------------------------
# Losses  (metrics/losses.py) and regularisers (models/regularization.py)
# --------------------------------------------------------------------------------------
def loss_l2_half(out: Tensor, gt: Tensor) -> Tensor:
    """0.5 * torch.nn.MSELoss() (train.py:82,182)."""
    return 0.5 * torch.mean((out - gt) ** 2)


def loss_l1_half(out: Tensor, gt: Tensor) -> Tensor:
    """0.5 * torch.nn.L1Loss() (train.py:92,182)."""
    return 0.5 * torch.mean(torch.abs(out - gt))


def loss_hdr(out: Tensor, gt: Tensor, kcoords: Tensor, opts: dict) -> Tuple[Tensor, Tensor]:
    """HDRLoss_FF.forward (losses.py:236-264) in its separable O(B) form.  The reference's
    ``input * filter_value`` broadcasts [M] x [B,1] -> [B,M], whose mean equals
    factor * mean_i((1-f_i)^2) * mean_j(|y_j|^2/den_j^2) exactly; value and gradient are checked
    against the reference class in tests/test_oracle_golden.py."""
    sigma, eps, factor = float(opts["hdr_ff_sigma"]), float(opts["hdr_eps"]), float(opts["hdr_ff_factor"])
    d2 = kcoords[..., 1] ** 2 + kcoords[..., 2] ** 2
    f = torch.exp(-d2 / (2 * sigma ** 2))
    y = torch.view_as_complex(out.contiguous())
    t = torch.view_as_complex(gt.contiguous())
    den = y.detach().abs() + eps
    err = (y - t).abs()
    loss = torch.log(err / den) ** 2
    A = torch.mean((1.0 - f) ** 2)
    reg = factor * A * torch.mean((y.abs() / den) ** 2)
    return loss.mean() + reg, reg


def loss_center(out: Tensor, gt: Tensor, kcoords: Tensor, opts: dict) -> Tuple[Tensor, int]:
    """CenterLoss.forward (losses.py:141-201; 'LSL' of train.py:87-88).  Pointwise part: 0.1 error_loss.mean() +
    0.9 (abs_loss.mean() + reg.mean()) with error_loss == abs_loss == (|y - t| / (|y|_detached + eps))^2 (:160-169) and
    reg the same [B,1] x [B] -> [B,B] broadcast as HDRLoss_FF (:170-171; separable form as in loss_hdr).  Pair term
    (:173-199): N_BANDS = 2 radial bands on dist^2 = ky^2 + kx^2 compared with the band RATIOS (0.1 | 0.5, then 0.5 | 1.0);
    n = min(min_sample, |inner|, |ring|) pairs per band drawn with torch.randperm on the default (CPU) generator, inner
    first; 0.1 * sum_bands mean((diff_gt - diff_pred)^2)."""
    sigma, eps, factor = float(opts["hdr_ff_sigma"]), float(opts["hdr_eps"]), float(opts["hdr_ff_factor"])
    min_sample = int(opts["min_sample"])
    d2 = kcoords[..., 1] ** 2 + kcoords[..., 2] ** 2
    f = torch.exp(-d2 / (2 * sigma ** 2))
    y = torch.view_as_complex(out.contiguous())
    t = torch.view_as_complex(gt.contiguous())
    den = y.detach().abs() + eps
    rel = ((y - t).abs() / den) ** 2
    reg = factor * torch.mean((1.0 - f) ** 2) * torch.mean((y.abs() / den) ** 2)
    ya, ta = y.abs(), t.abs()
    center = torch.zeros((), dtype=out.dtype, device=out.device)
    n_bands = 2
    for band in range(1, n_bands + 1):
        r1 = (band - 1) / n_bands
        if r1 == 0:
            r1 = 0.1
        m1 = d2 <= r1
        m2 = (d2 <= band / n_bands) & ~m1
        y1, y2 = ya[m1], ya[m2]
        n = min(min_sample, min(len(y1), len(y2)))
        if n == 0:
            continue
        a = torch.randperm(y1.size(0))[:n]
        b = torch.randperm(y2.size(0))[:n]
        diff_pred = y1[a] - y2[b]
        diff_gt = ta[m1][a] - ta[m2][b]
        center = center + ((diff_gt - diff_pred) ** 2).mean()
    return 0.1 * rel.mean() + 0.9 * (rel.mean() + reg) + 0.1 * center, 0


def loss_tanh(out: Tensor, gt: Tensor) -> Tuple[Tensor, float]:
    """TanhL2Loss.forward, with_mag=False (losses.py:130-139)."""
    return torch.mean((torch.tanh(out) - torch.tanh(gt)) ** 2), 0


def loss_logspace(out: Tensor, gt: Tensor, opts: dict) -> Tensor:
    """LogSpaceLoss.forward (losses.py:214-223)."""
    eps = float(opts["hdr_eps"])
    y = torch.view_as_complex(out.contiguous())
    t = torch.view_as_complex(gt.contiguous())
    return torch.mean(((y - t).abs() / (y.detach().abs() + eps)) ** 2)


def loss_msle(out: Tensor, gt: Tensor, eps: float = 1e-9) -> Tensor:
    """MSLELoss.forward (losses.py:25-27)."""
    return torch.mean((torch.log(out + 1 + eps) - torch.log(gt + 1 + eps)) ** 2)


def loss_consistency(outs: Sequence[Tensor], dist: Tensor, bounds) -> Tensor:
    """ConsistencyLoss.forward (losses.py:315-324).  ``dist`` [B] selects whole rows; ``dist``
    [B,1] (per-coil mode) makes torch.where return (rows, zeros) so only channel 0 is compared
     -- reproduced by using the same indexing."""
    loss = 0
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i]
        ind = torch.where((dist < lo) | (dist > hi))
        if ind[0].numel():
            loss = loss + torch.mean((outs[i][ind].detach() - outs[i + 1][ind]) ** 2)
    return loss


def loss_tv(img: Tensor, weight: float = 0.0001) -> Tensor:
    """tv_loss (losses.py:326-343) on one coil grid [H,W,2]."""
    w_var = torch.mean(torch.abs(img[:, :-1, :] - img[:, 1:, :]))
    h_var = torch.mean(torch.abs(img[:-1, :, :] - img[1:, :, :]))
    return weight * (h_var + w_var)


def reg_l1(params: Sequence[Tensor], strength: float) -> Tensor:
    """Regularization_L1.__call__ (regularization.py:25-28)."""
    return sum(torch.sum(torch.abs(p)) for p in params) * strength


def reg_l2(params: Sequence[Tensor], strength: float) -> Tensor:
    """Regularization_L2.__call__ (regularization.py:34-36)."""
    return abs(sum(torch.sum(p.pow(2)) for p in params)) * strength


def make_loss(config: dict):
    """Loss selection of train.py:81-98,178-182 -> callable(out, gt, kcoords) -> scalar."""
    kind = config["loss"]
    opts = config.get("loss_opts", {})
    if kind == "L2":
        return lambda o, g, k: loss_l2_half(o, g)
    if kind == "L1":
        return lambda o, g, k: loss_l1_half(o, g)
    if kind == "MSLE":
        return lambda o, g, k: 0.5 * loss_msle(o, g)
    if kind == "HDR":
        return lambda o, g, k: loss_hdr(o, g, k, opts)[0]
    if kind == "tanh":
        return lambda o, g, k: loss_tanh(o, g)[0]
    if kind == "LSL":
        return lambda o, g, k: loss_center(o, g, k, opts)[0]
    raise NotImplementedError(kind)


# --------------------------------------------------------------------------------------
# Adam + LambdaLR as used (train.py:76,153,251).  torch.optim.Adam is third-party;
# this is its published single-tensor algorithm (amsgrad=False, maximize=False, eps=1e-8).
# --------------------------------------------------------------------------------------
def adam_init(params: Dict[str, Tensor]):
    return {k: {"step": 0, "m": torch.zeros_like(torch.view_as_real(p) if p.is_complex() else p),
                "v": torch.zeros_like(torch.view_as_real(p) if p.is_complex() else p)} for k, p in params.items()}


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Optional[Tensor]], state, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0):
    """In-place Adam update; parameters whose grad is None are skipped (torch semantics).
    Complex parameters are updated as two independent reals (torch views them as real)."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        pr = torch.view_as_real(p) if p.is_complex() else p
        gr = torch.view_as_real(g.contiguous()) if g.is_complex() else g
        st = state[k]
        st["step"] += 1
        t = st["step"]
        if weight_decay != 0:
            gr = gr + weight_decay * pr
        st["m"].mul_(beta1).add_(gr, alpha=1 - beta1)
        st["v"].mul_(beta2).addcmul_(gr, gr, value=1 - beta2)
        bc1 = 1 - beta1 ** t
        bc2 = 1 - beta2 ** t
        step_size = lr / bc1
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        pr.addcdiv_(st["m"], denom, value=-step_size)


def lr_factor(epoch: int, max_epoch: int) -> float:
    """LambdaLR(optim, lambda x: 0.2**min(x/max_epoch, 1)) (train.py:153)."""
    return 0.2 ** min(epoch / max_epoch, 1)


# --------------------------------------------------------------------------------------
# Eval chain (train.py:221-231; models/utils.py:236-250).  fastmri is third-party and absent:
# these follow fastmri 0.3.0's published definitions (centred orthonormal FFT over dims (-3,-2)
# of a (...,2) real view; complex_abs; root-sum-of-squares).  Parity unpinned for this stage.
# --------------------------------------------------------------------------------------
def complex_abs(x: Tensor) -> Tensor:
    return (x ** 2).sum(dim=-1).sqrt()


def rss(x: Tensor, dim: int = 0) -> Tensor:
    return torch.sqrt((x ** 2).sum(dim))


def _fftc(x: Tensor, inverse: bool) -> Tensor:
    c = torch.view_as_complex(x.contiguous())
    c = torch.fft.ifftshift(c, dim=(-2, -1))
    c = (torch.fft.ifftn if inverse else torch.fft.fftn)(c, dim=(-2, -1), norm="ortho")
    c = torch.fft.fftshift(c, dim=(-2, -1))
    return torch.view_as_real(c)


def fft2c(x: Tensor) -> Tensor:
    return _fftc(x, False)


def ifft2c(x: Tensor) -> Tensor:
    return _fftc(x, True)


def psnr(x: Tensor, xhat: Tensor, epsilon: float = 1e-10) -> Tensor:
    """models/utils.py:236-250 -- note max(x), not max(x)^2."""
    denom = torch.mean((x - xhat) ** 2)
    return 10 * torch.log10(torch.max(x) / (denom + epsilon))


def reconstruct(flat: Tensor, shape, in_image_space: bool) -> Tensor:
    """train.py:221-229: [(C*H*W),2] -> (C,H,W,2) -> (ifft2c) -> abs -> rss over coils."""
    C, H, W = shape
    im = flat.reshape(C, H, W, 2)
    if not in_image_space:
        im = ifft2c(im)
    return rss(complex_abs(im), dim=0)


# --------------------------------------------------------------------------------------
# Training loops (train.py:155-198; train_kspace_multiscale.py:161-201)
# --------------------------------------------------------------------------------------
class PerItemDataset(torch.utils.data.Dataset):
    """The reference's data path for one batch: MRIDataset.__getitem__ returns ONE coordinate's
    (coords[idx], image[idx], [], []) (nerp_datasets.py:236-237) and the loops draw batches of `batch_size` of them from
    a torch DataLoader with the default collate, shuffle=False, num_workers=0 (models/utils.py:84-90) -- 25 000 Python calls
    and a 25 000-way stack per batch.  Used by bench.py to time the host-bound figure the reference would actually see."""

    def __init__(self, coords: Tensor, image: Tensor):
        self.coords, self.image = coords, image

    def __getitem__(self, idx):
        return self.coords[idx], self.image[idx], list(), list()

    def __len__(self):
        return len(self.image)


def _batches(n: int, bs: int):
    """Sequential, unshuffled, drop_last=False (models/utils.py:84-90)."""
    for lo in range(0, n, bs):
        yield lo, min(lo + bs, n)


def train_single_scale(config: dict, sd, enc_B, coords: Tensor, image: Tensor, max_steps: int,
                       mask: Optional[Tensor] = None, record=None, grid_hw=None):
    """The per-step loop of train.py:158-192 on pre-built tensors (the DataLoader contract of
    SURVEY 3.1: batch i = rows [i*bs,(i+1)*bs) of the C-major flattened grid).  Mutates ``sd``.
    ``mask`` [N] bool = undersampling mask (forward on all rows, loss on masked rows,
    train.py:172-177).  ``config['per_coil']`` with ``grid_hw=(H, W)``: one coil per step
    (MRICoilWrapperDataset, nerp_datasets.py:397-441; loader batch_size 1, models/utils.py:65-66),
    plus tv_loss on the coil grid when ``config['use_tv']`` and a mask is given (train.py:173-175).
    Returns the list of per-step loss values."""
    model = config["model"]
    net = config["net"]
    keys = trainable_keys(model, sd)
    params = {k: sd[k].requires_grad_(True) for k in keys}
    state = adam_init(params)
    loss_fn = make_loss(config)
    reg = config.get("regularization", {"type": "none"})
    n, bs = coords.shape[0], config["batch_size"]
    if config.get("per_coil", False):
        bs = grid_hw[0] * grid_hw[1]
    losses, step = [], 0
    for epoch in range(config["max_epoch"]):  # train.py:155 / train_kspace_multiscale.py:161
        if step >= max_steps:
            break
        lr = config["lr"] * lr_factor(epoch, config["max_epoch"])
        for lo, hi in _batches(n, bs):
            if step >= max_steps:
                break
            kc, gt = coords[lo:hi], image[lo:hi]
            x = encode(kc, enc_B, config["encoder"]["embedding"])
            out = model_forward(model, sd, x, net)
            tv = None
            if mask is not None:
                if config.get("use_tv", False):
                    tv = loss_tv(out.view(grid_hw[0], grid_hw[1], 2))
                m = mask[lo:hi]
                out, gt = out[m], gt[m]
            loss = loss_fn(out, gt, kc)
            if tv is not None:
                loss = tv + loss
            # regularization(model.parameters()) (train.py:185-187) runs over EVERY Parameter, the frozen omega_0 / scale_0 of
            # the WIRE layers included (networks.py:191-192, wire2d.py:35-36): they add a constant to the L1 value and sit
            # inside the modulus of the L2 value
            if reg["type"] == "L1":
                loss = loss + reg_l1(list(sd.values()), reg["strenght"])
            elif reg["type"] == "L2":
                loss = loss + reg_l2(list(sd.values()), reg["strenght"])
            grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
            with torch.no_grad():
                adam_step({k: p for k, p in params.items()}, dict(zip(keys, grads)), state, lr,
                          config["beta1"], config["beta2"], 1e-8, config["weight_decay"])
            losses.append(float(loss.detach()))
            step += 1
            if record is not None:
                record(step, sd, float(loss.detach()))
    for k in keys:
        sd[k].requires_grad_(False)
    return losses


def create_pairs(radii, n: int = 1):
    """train_kspace_multiscale.py:42-47 -- nested discs (r_0, r_k), each repeated n times."""
    pairs = []
    for r in radii[1:]:
        pairs += [(radii[0], r)] * n
    return pairs


def train_multiscale(config: dict, sd, enc_B, coords: Tensor, image: Tensor, dist: Tensor, radii,
                     max_steps: int, record=None, mask: Optional[Tensor] = None, grid_hw=None):
    """The loop of train_kspace_multiscale.py:164-195 for model in {MultiscaleKFourier, BoundedFourier}; loss in
    {L2, L1, LSL->LogSpaceLoss}.  ``mask`` [N] bool: undersampling (:176-182: the pointwise terms see the sampled
    rows, ConsistencyLoss and TV all rows); ``config['per_coil']`` with ``grid_hw``: one coil per step, plus
    tv_loss on the last head when ``config['use_tv']`` (:173-175)."""
    model = config["model"]
    model = {"Fourier": "MultiscaleKFourier"}.get(model, model)
    net = config["net"]
    pairs = create_pairs(radii, 1)
    pairs_model = create_pairs(radii, 2)
    keys = trainable_keys(model, sd)
    params = {k: sd[k].requires_grad_(True) for k in keys}
    state = adam_init(params)
    opts = config.get("loss_opts", {})
    kind = config["loss"]
    n, bs = coords.shape[0], config["batch_size"]
    if config.get("per_coil", False):
        bs = grid_hw[0] * grid_hw[1]
    losses, step = [], 0
    for epoch in range(config["max_epoch"]):  # train.py:155 / train_kspace_multiscale.py:161
        if step >= max_steps:
            break
        lr = config["lr"] * lr_factor(epoch, config["max_epoch"])
        for lo, hi in _batches(n, bs):
            if step >= max_steps:
                break
            kc, gt, d = coords[lo:hi], image[lo:hi], dist[lo:hi]
            x = encode(kc, enc_B, config["encoder"]["embedding"])
            outs = model_forward(model, sd, x, net, dist_to_center=d, boundaries=pairs_model)
            loss = 0.1 * loss_consistency(outs, d, pairs)
            if config.get("use_tv", False):
                loss = loss + loss_tv(outs[-1].view(grid_hw[0], grid_hw[1], 2))
            if mask is not None:
                m = mask[lo:hi]
                gt = gt[m]
            for out in outs:  # limit_kspace is a no-op: every head sees the full gt
                if mask is not None:
                    out = out[m]
                if kind == "L2":
                    loss = loss + loss_l2_half(out, gt)
                elif kind == "L1":
                    loss = loss + loss_l1_half(out, gt)
                elif kind == "LSL":
                    loss = loss + 0.5 * loss_logspace(out, gt, opts)
                else:
                    raise NotImplementedError(kind)
            grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
            with torch.no_grad():
                adam_step(params, dict(zip(keys, grads)), state, lr,
                          config["beta1"], config["beta2"], 1e-8, config["weight_decay"])
            losses.append(float(loss.detach()))
            step += 1
            if record is not None:
                record(step, sd, float(loss.detach()))
    for k in keys:
        sd[k].requires_grad_(False)
    return losses
