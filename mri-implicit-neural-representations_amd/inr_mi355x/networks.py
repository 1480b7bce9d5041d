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
from .engine import MLPEngine, encode_gauss

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
            # not on any BASELINE config; kept as plain tensor ops on the device (no HIP kernel yet)
            parts = []
            for a in range(3):
                p = (2.0 * np.pi * x[:, a:a + 1]) @ self.B.T
                parts.append(torch.cat((torch.sin(p), torch.cos(p)), dim=-1))
            return torch.cat(parts, dim=-1)
        return x


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
        grads = [flat_grad[o:o + n].view(s) for (o, n, s) in module._layout]
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


class _FlatMLP(nn.Module):
    """Common machinery: flat parameter buffer + views + engine."""

    _kind = L.KIND_SIREN

    def _build(self, dims: List[int], init_fn, key_modules, last_act: int):
        """dims = [in, w, ..., w, out].  init_fn(k, in_f, out_f) -> (weight, bias) consumes the RNG
        in the reference's order."""
        self._dims = dims
        self._last_act = last_act
        tensors = [init_fn(k, dims[k], dims[k + 1]) for k in range(len(dims) - 1)]
        P = sum(w.numel() + b.numel() for w, b in tensors)
        flat = torch.empty(P)
        self._layout = []
        off = 0
        for w, b in tensors:
            for t in (w, b):
                flat[off:off + t.numel()] = t.reshape(-1)
                self._layout.append((off, t.numel(), tuple(t.shape)))
                off += t.numel()
        self._flat = flat
        self._eng: Optional[MLPEngine] = None
        holders = []
        for k in range(len(dims) - 1):
            ow, nw, sw = self._layout[2 * k]
            ob, nb_, sb = self._layout[2 * k + 1]
            holders.append(_Holder(nn.Parameter(flat[ow:ow + nw].view(sw)), nn.Parameter(flat[ob:ob + nb_].view(sb))))
        self.model = key_modules(holders)

    # moving the module re-points every parameter at a view of the moved flat buffer
    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        params = list(self.parameters())
        for p, (o, n, s) in zip(params, self._layout):
            p.data = new_flat[o:o + n].view(s)
            if p.grad is not None:
                p.grad = None
        self._flat = new_flat
        self._eng = None
        return self

    def _engine(self) -> MLPEngine:
        if not self._flat.is_cuda:
            raise RuntimeError("inr_mi355x models run on an MI355X only: call .to('cuda') first (no CPU fallback)")
        if self._eng is None:
            d = self._dims
            self._eng = MLPEngine(self._kind, d[0], d[1], len(d) - 1, d[-1], self._last_act, L.INPUT_X, 0, SIREN_W0)
            self._eng.bind(self._flat)
        return self._eng

    def fused_engine(self, enc_size: int) -> MLPEngine:
        """Tier-2 engine over the SAME flat parameters with the gauss encoder fused into layer 0."""
        d = self._dims
        eng = MLPEngine(self._kind, d[0], d[1], len(d) - 1, d[-1], self._last_act, L.INPUT_GAUSS, enc_size, SIREN_W0)
        eng.bind(self._flat)
        return eng

    def forward(self, x):
        return _MLPFunction.apply(self, x, *self.parameters())


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
