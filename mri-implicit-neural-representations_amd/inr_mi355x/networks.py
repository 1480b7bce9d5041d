"""Drop-in mirrors of the reference's model classes for the MI355X engine.

Same constructor signatures (``Model(params: dict)``), same ``state_dict`` keys / shapes / order,
same RNG consumption at construction (so ``torch.manual_seed(s)`` gives the reference's initial
weights bit for bit), same ``forward`` contract -- but all parameters are views into ONE flat fp32
buffer and ``forward``/``backward`` run the hand-written gfx950 kernels through the C-ABI.

Reference: models/networks.py  Positional_Encoder :7-35, FFN :48-69, SirenLayer :74-96, SIREN :99-124.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .engine import encode_logf, MLPEngine, encode_gauss

SIREN_W0 = 30.0  # SirenLayer(w0=30) for every layer, first included (networks.py:75,114-117)


class Positional_Encoder:
    """networks.py:7-35.  ``B`` is sampled on the CPU generator first and only then moved
    (networks.py:13-14), exactly like the reference."""

    def __init__(self, params, device):
        self.device = device
        self.B = None
        self.embedding_type = params["embedding"]
        if params["embedding"] == "gauss":
            self.B = torch.randn((params["embedding_size"], params["coordinates_size"])) * params["scale"]
            self.B = self.B.to(device)
        elif params["embedding"] == "LogF":
            self.B = 2.0 ** torch.linspace(0.0, params["scale"], steps=int(params["embedding_size"] / (2 * params["coordinates_size"]))).reshape(-1, 1)
            self.B = self.B.to(device)
        elif params["embedding"] == "none":
            pass
        else:
            raise NotImplementedError

    def embedding(self, x):
        if self.embedding_type == "gauss":
            return encode_gauss(x.contiguous(), self.B.contiguous())
        if self.embedding_type == "LogF":
            return encode_logf(x.contiguous(), self.B.contiguous())
        return x


def _view(flat: torch.Tensor, off: int, n: int, shape, is_complex: bool) -> torch.Tensor:
    """View of the flat fp32 buffer as the parameter tensor (complex64 = interleaved (re, im) pairs)."""
    if is_complex:
        return torch.view_as_complex(flat[off:off + n].view(*shape, 2))
    return flat[off:off + n].view(shape)


class _MLPFunction(torch.autograd.Function):
    """Tier-1 bridge: ``model(x)`` + ``loss.backward()`` + stock ``torch.optim.Adam`` work unchanged."""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module._engine()
        eng.pack()  # parameters may have been stepped by an external optimizer
        need_grad = any(ctx.needs_input_grad[2:])  # (grad mode is always off inside Function.forward)
        x = x.contiguous()
        out = eng.forward(x, None, save=need_grad)
        ctx.module = module
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        module = ctx.module
        eng = module._engine()
        flat_grad = eng.backward(x, None, dout.contiguous())
        # clones: the engine's gradient buffer is reused by the next backward, and AccumulateGrad may adopt what it
        # is handed as .grad (a second backward without zero_grad must accumulate, not overwrite)
        grads = [_view(flat_grad, o, n, s, c).clone() for (o, n, s, c) in module._layout]
        return (None, None, *grads)


class _Holder(nn.Module):
    """A module that only holds a weight and a bias (views into the flat buffer)."""

    def __init__(self, weight: nn.Parameter, bias: nn.Parameter):
        super().__init__()
        self.weight = weight
        self.bias = bias


class _SirenLayerShell(nn.Module):
    def __init__(self, holder: _Holder):
        super().__init__()
        self.linear = holder  # key: model.{k}.linear.weight / .bias


class _FlatModel(nn.Module):
    """Common machinery: ONE flat fp32 parameter buffer, nn.Parameters that are views of it (in
    state_dict order), and the engine bound to it."""

    def _flatten(self, tensors: List[torch.Tensor]) -> List[nn.Parameter]:
        """tensors: the trainable tensors in state_dict order (real or complex64)."""
        reals = [torch.view_as_real(t) if t.is_complex() else t for t in tensors]
        P = sum(r.numel() for r in reals)
        flat = torch.empty(P)
        self._layout = []
        off = 0
        for t, r in zip(tensors, reals):
            flat[off:off + r.numel()] = r.reshape(-1)
            self._layout.append((off, r.numel(), tuple(t.shape), t.is_complex()))
            off += r.numel()
        self._flat = flat
        self._eng: Optional[MLPEngine] = None
        self._flat_params = [nn.Parameter(_view(flat, o, n, s, c)) for (o, n, s, c) in self._layout]
        return self._flat_params

    # moving the module re-points every parameter at a view of the moved flat buffer
    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        for p, (o, n, s, c) in zip(self._flat_params, self._layout):
            p.data = _view(new_flat, o, n, s, c)
            p.grad = None
        for p in self.parameters():  # frozen extras (WIRE's omega_0 / scale_0) are ordinary tensors
            if not any(p is q for q in self._flat_params):
                p.data = fn(p.data)
        self._flat = new_flat
        self._eng = None
        return self

    def _make_engine(self, input_mode: int, enc_size: int) -> MLPEngine:
        raise NotImplementedError

    def _engine(self) -> MLPEngine:
        if not self._flat.is_cuda:
            raise RuntimeError("inr_mi355x models run on an MI355X only: call .to('cuda') first (no CPU fallback)")
        if self._eng is None:
            self._eng = self._make_engine(L.INPUT_X, 0)
            self._eng.bind(self._flat)
        return self._eng

    def fused_engine(self, enc_size: int, precision: str = "f32") -> MLPEngine:
        """Tier-2 engine over the SAME flat parameters with the gauss encoder fused into layer 0.
        ``precision="bf16"``: the bf16-MFMA throughput path (SIREN, width 129..256; fp32 master weights)."""
        if not self._flat.is_cuda:
            raise RuntimeError("inr_mi355x models run on an MI355X only: call .to('cuda') first (no CPU fallback)")
        if precision not in ("f32", "bf16"):
            raise ValueError(f"precision {precision!r}")
        if precision == "bf16":
            eng = self._make_engine(L.INPUT_GAUSS, enc_size, L.PRECISION_BF16)
        else:
            eng = self._make_engine(L.INPUT_GAUSS, enc_size)
        eng.bind(self._flat)
        return eng

    def forward(self, x):
        return _MLPFunction.apply(self, x, *self._flat_params)


class _FlatMLP(_FlatModel):
    _kind = L.KIND_SIREN

    def _build(self, dims: List[int], init_fn, key_modules, last_act: int):
        """dims = [in, w, ..., w, out].  init_fn(k, in_f, out_f) -> (weight, bias) consumes the RNG
        in the reference's order."""
        self._dims = dims
        self._last_act = last_act
        tensors = []
        for k in range(len(dims) - 1):
            tensors += list(init_fn(k, dims[k], dims[k + 1]))
        ps = self._flatten(tensors)
        holders = [_Holder(ps[2 * k], ps[2 * k + 1]) for k in range(len(dims) - 1)]
        self.model = key_modules(holders)

    def _make_engine(self, input_mode: int, enc_size: int, precision: int = L.PRECISION_F32) -> MLPEngine:
        d = self._dims
        return MLPEngine(self._kind, d[0], d[1], len(d) - 1, d[-1], self._last_act, input_mode, enc_size, SIREN_W0,
                         precision=precision)


class SIREN(_FlatMLP):
    """networks.py:99-124 (+ SirenLayer :74-96)."""

    _kind = L.KIND_SIREN

    def __init__(self, params):
        super().__init__()
        num_layers = params["network_depth"]
        hidden_dim = params["network_width"]
        input_dim = params["network_input_size"]
        output_dim = params["network_output_size"]
        last_linear = params.get("network_last_linear", True)
        last_tanh = params.get("last_tanh", False)
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]

        def init_fn(k, in_f, out_f):
            lin = nn.Linear(in_f, out_f)  # default init first (weight, then bias), networks.py:79
            b = 1 / in_f if k == 0 else np.sqrt(6 / in_f) / SIREN_W0  # init_weights, networks.py:85-89
            with torch.no_grad():
                lin.weight.uniform_(-b, b)
            return lin.weight.detach(), lin.bias.detach()

        if last_tanh:
            last_act = L.ACT_TANH
        elif last_linear:
            last_act = L.ACT_ID
        else:
            last_act = L.ACT_SIN
        self._build(dims, init_fn, lambda hs: nn.Sequential(*[_SirenLayerShell(h) for h in hs]), last_act)


class FFN(_FlatMLP):
    """networks.py:48-69: ReLU hidden layers, Sigmoid output; keys model.{0,2,4,...}.weight/bias."""

    _kind = L.KIND_FFN

    def __init__(self, params):
        super().__init__()
        num_layers = params["network_depth"]
        hidden_dim = params["network_width"]
        dims = [params["network_input_size"]] + [hidden_dim] * (num_layers - 1) + [params["network_output_size"]]

        def init_fn(k, in_f, out_f):
            lin = nn.Linear(in_f, out_f)
            return lin.weight.detach(), lin.bias.detach()

        def seq(holders):
            mods = []
            for h in holders:
                mods += [h, nn.Identity()]  # activations occupy the odd Sequential slots
            return nn.Sequential(*mods)

        self._build(dims, init_fn, seq, L.ACT_SIGMOID)


class _GaborLayerShell(nn.Module):
    """ComplexGaborLayer's parameter set (networks.py:191-197): frozen omega_0, scale_0, then linear."""

    def __init__(self, omega0: float, sigma0: float, holder: _Holder):
        super().__init__()
        self.omega_0 = nn.Parameter(omega0 * torch.ones(1), False)
        self.scale_0 = nn.Parameter(sigma0 * torch.ones(1), False)
        self.linear = holder


class WIRE(_FlatModel):
    """networks.py:206-260.  Complex64 layers; ``hidden = int(network_width / sqrt(2))`` (:228);
    ``network_depth`` counts the hidden complex layers (:241-245); output is the real part (:257-258)
    -- returned here as a contiguous [B,2] tensor (the reference returns the strided ``.real`` view)."""

    def __init__(self, params):
        super().__init__()
        self.hidden_layers = params["network_depth"]
        in_features = params["network_input_size"]
        out_features = params["network_output_size"]
        self.first_omega_0 = params["first_omega_0"]
        self.hidden_omega_0 = params["hidden_omega_0"]
        self.scale = params["scale"]
        self.hidden_features = int(params["network_width"] / np.sqrt(2))
        self.in_features, self.out_features = in_features, out_features
        hid = self.hidden_features
        tensors = []
        lin = nn.Linear(in_features, hid, dtype=torch.float)  # is_first: real weights (networks.py:185-197)
        tensors += [lin.weight.detach(), lin.bias.detach()]
        for _ in range(self.hidden_layers):
            lin = nn.Linear(hid, hid, dtype=torch.cfloat)
            tensors += [lin.weight.detach(), lin.bias.detach()]
        lin = nn.Linear(hid, out_features, dtype=torch.cfloat)  # final_linear (networks.py:247-250)
        tensors += [lin.weight.detach(), lin.bias.detach()]
        ps = self._flatten(tensors)
        mods = [_GaborLayerShell(self.first_omega_0, self.scale, _Holder(ps[0], ps[1]))]
        for k in range(self.hidden_layers):
            mods.append(_GaborLayerShell(self.hidden_omega_0, self.scale, _Holder(ps[2 + 2 * k], ps[3 + 2 * k])))
        mods.append(_Holder(ps[-2], ps[-1]))
        self.net = nn.Sequential(*mods)

    def _make_engine(self, input_mode: int, enc_size: int) -> MLPEngine:
        if input_mode != L.INPUT_X:
            raise NotImplementedError("WIRE takes raw coordinates (encoder.embedding: none)")
        return MLPEngine(L.KIND_WIRE, self.in_features, self.hidden_features, self.hidden_layers, self.out_features,
                         L.ACT_ID, L.INPUT_X, 0, 0.0, float(self.first_omega_0), float(self.hidden_omega_0),
                         float(self.scale))


class _GaborLayer2DShell(nn.Module):
    """net.k of WIRE2D: frozen omega_0 / scale_0, then linear and scale_orth (wire2d.py:36-47)."""

    def __init__(self, omega0: float, sigma0: float, lin: _Holder, orth: _Holder):
        super().__init__()
        self.omega_0 = nn.Parameter(omega0 * torch.ones(1), False)
        self.scale_0 = nn.Parameter(sigma0 * torch.ones(1), False)
        self.linear = lin
        self.scale_orth = orth


class WIRE2D(_FlatModel):
    """wire2d.py:62-117.  ComplexGaborLayer2D (:3-60): y = exp(j w0 lin) * exp(-s0^2 (|lin|^2 + |orth|^2)) with a
    second Linear ``scale_orth`` per layer; the hidden width is NOT reduced (:76); output is the real part.
    ``last_tanh`` puts ``torch.nn.Tanh()`` on the complex output before ``.real`` (:106-107, :113-117)."""

    def __init__(self, params):
        super().__init__()
        self.last_tanh = bool(params.get("last_tanh", False))
        self.hidden_layers = params["network_depth"]
        self.hidden_features = params["network_width"]
        self.in_features = params["network_input_size"]
        self.out_features = params["network_output_size"]
        self.first_omega_0 = params["first_omega_0"]
        self.hidden_omega_0 = params["hidden_omega_0"]
        self.scale = params["scale"]
        hid = self.hidden_features
        tensors = []
        for k in range(self.hidden_layers + 1):  # linear, then scale_orth: the RNG order of wire2d.py:40-47
            fin, dt = (self.in_features, torch.float) if k == 0 else (hid, torch.cfloat)
            lin = nn.Linear(fin, hid, dtype=dt)
            orth = nn.Linear(fin, hid, dtype=dt)
            tensors += [lin.weight.detach(), lin.bias.detach(), orth.weight.detach(), orth.bias.detach()]
        fin = nn.Linear(hid, self.out_features, dtype=torch.cfloat)
        tensors += [fin.weight.detach(), fin.bias.detach()]
        ps = self._flatten(tensors)
        mods = []
        for k in range(self.hidden_layers + 1):
            om = self.first_omega_0 if k == 0 else self.hidden_omega_0
            mods.append(_GaborLayer2DShell(om, self.scale, _Holder(ps[4 * k], ps[4 * k + 1]),
                                           _Holder(ps[4 * k + 2], ps[4 * k + 3])))
        mods.append(_Holder(ps[-2], ps[-1]))
        self.net = nn.Sequential(*mods)

    def _make_engine(self, input_mode: int, enc_size: int) -> MLPEngine:
        if input_mode != L.INPUT_X:
            raise NotImplementedError("WIRE2D takes raw coordinates (encoder.embedding: none)")
        return MLPEngine(L.KIND_WIRE2D, self.in_features, self.hidden_features, self.hidden_layers, self.out_features,
                         L.ACT_CTANH if self.last_tanh else L.ACT_ID, L.INPUT_X, 0, 0.0, float(self.first_omega_0),
                         float(self.hidden_omega_0), float(self.scale))
