"""Drop-in mirrors of the reference's multiplicative filter networks (models/mfn.py):
``FourierNet`` (:61-94), ``GaborNet`` / ``KGaborNet`` (:133-204), ``MultiscaleKFourier`` (:206-267) and
``MultiscaleBoundedFourier`` (:288-355).  Same constructor signature, RNG
order, state_dict keys / order (``linear.*``, ``output_linear(.k).*``, ``filters.k.linear.*``).

Two ways to call a model, both on the hand-written kernels:

  * the reference's own contract (train.py:163-169, mfn.py:34-43,85-94,255-267): ``model(x)`` on the ENCODED
    input ``x = encoder.embedding(coords)`` [B, network_input_size], any encoder ('gauss' | 'LogF' | 'none'):

        enc = Positional_Encoder(config['encoder'], device)      # consumes the RNG first, like train.py:52
        model = MultiscaleKFourier(config['net']).to(device)
        outs = model(coords=enc.embedding(coords), dist_to_center=dist)   # list of 4 [B,2] tensors (heads 1,3,5,7)

  * fused: after ``model.bind_encoder(enc)`` ('gauss' only) the model may also be called on RAW coordinates
    [B,3]; the kernel then generates the features itself and the [B,2E] tensor is never materialised.  The
    trainers use this form (``fused_engine``).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from .engine import MFNEngine
from .networks import _FlatModel, _Holder, _view


class _MFNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, dist, *params):
        mode = module._mode_of(x)
        eng = module._engine(mode)
        eng.pack()
        need_grad = any(ctx.needs_input_grad[3:])
        x = x.contiguous()
        dist = dist.reshape(-1).contiguous() if dist is not None else None
        enc_B = module._enc_B if mode == "gauss" else None
        out = eng.forward(x, enc_B, save=need_grad, dist=dist)
        ctx.module, ctx.dist, ctx.mode = module, dist, mode
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        module = ctx.module
        eng = module._engine(ctx.mode)
        enc_B = module._enc_B if ctx.mode == "gauss" else None
        flat_grad = eng.backward(x, enc_B, dout.contiguous(), dist=ctx.dist)
        grads = []
        for (o, n, s, c), live in zip(module._layout, module._live):
            # clones: the engine's gradient buffer is reused by the next backward, and AccumulateGrad may adopt
            # what it is handed as .grad; dead layers: grad None, like autograd
            grads.append(_view(flat_grad, o, n, s, c).clone() if live else None)
        return (None, None, None, *grads)


class _FilterShell(nn.Module):
    def __init__(self, holder: _Holder):
        super().__init__()
        self.linear = holder  # filters.k.linear.weight / bias


class _GaborShell(nn.Module):
    """filters.k of GaborNet: own parameters mu, gamma precede the child Linear in state_dict order
    (mfn.py:101-111)."""

    def __init__(self, mu: nn.Parameter, gamma: nn.Parameter, holder: _Holder):
        super().__init__()
        self.mu = mu
        self.gamma = gamma
        self.linear = holder


class _BoundedShell(nn.Module):
    def __init__(self, holder: _Holder, bounds):
        super().__init__()
        self.linear = holder  # linear.k.linear.weight / bias (BoundedLinear wraps a Linear, mfn.py:277-279)
        self.bounds = bounds


class _MFNBase(_FlatModel):
    _multiscale = False
    _bounds = None
    _gabor = None  # (alpha, beta) of the GaborLayer gamma prior, or None for Fourier filters
    _output_act = False  # mfn.py:40-41: torch.sin on the output (single-head classes only)

    def _build_mfn(self, params, filter_scale: float, weight_scale: float):
        n = params["network_depth"]
        W = params["network_width"]
        in_size = params["network_input_size"]
        out_size = params["network_output_size"]
        self._n, self._W, self._in, self._out = n, W, in_size, out_size
        self._enc_B = None
        # MFNBase.__init__ (mfn.py:15-32): n hidden Linear, output Linear, then uniform_ on every hidden weight
        lin = [nn.Linear(W, W) for _ in range(n)]
        out_lin = nn.Linear(W, out_size)
        b = np.sqrt(weight_scale / W)
        for m in lin:
            m.weight.data.uniform_(-b, b)
        if self._bounds is not None:  # mfn.py:319-321: a fresh list of BoundedLinear (default init) replaces it
            lin = [nn.Linear(W, W) for _ in range(n)]
        # child: n+1 FourierLayer (mfn.py:50-55): default init, weight *= scale, bias ~ U(-pi, pi)
        filt, centres = [], []
        for _ in range(n + 1):
            m = nn.Linear(in_size, W)
            if self._gabor is not None:  # GaborLayer.__init__ (mfn.py:101-113): Linear, mu, gamma, scale, bias
                mu = 2 * torch.rand(W, in_size) - 1
                gamma = torch.distributions.gamma.Gamma(*self._gabor).sample((W,))
                m.weight.data *= filter_scale * torch.sqrt(gamma[:, None])
                centres.append((mu, gamma))
            else:
                m.weight.data *= filter_scale
            m.bias.data.uniform_(-np.pi, np.pi)
            filt.append(m)
        heads = [out_lin]
        if self._multiscale:  # mfn.py:247: output_linear replaced by n+1 fresh heads (after the filters)
            heads = [nn.Linear(W, out_size) for _ in range(n + 1)]
        tensors = []
        for m in lin + heads:  # state_dict order: linear, output_linear, filters
            tensors += [m.weight.detach(), m.bias.detach()]
        for i, m in enumerate(filt):
            if self._gabor is not None:
                tensors += list(centres[i])
            tensors += [m.weight.detach(), m.bias.detach()]
        ps = self._flatten(tensors)
        k = 0
        if self._bounds is not None:
            self.linear = nn.ModuleList([_BoundedShell(_Holder(ps[2 * i], ps[2 * i + 1]), self._bounds[i])
                                         for i in range(n)])
        else:
            self.linear = nn.ModuleList([_Holder(ps[2 * i], ps[2 * i + 1]) for i in range(n)])
        k = 2 * n
        if self._multiscale:
            self.output_linear = nn.ModuleList([_Holder(ps[k + 2 * i], ps[k + 2 * i + 1]) for i in range(n + 1)])
            k += 2 * (n + 1)
        else:
            self.output_linear = _Holder(ps[k], ps[k + 1])
            k += 2
        if self._gabor is not None:
            self.filters = nn.ModuleList([_GaborShell(ps[k + 4 * i], ps[k + 4 * i + 1],
                                                      _Holder(ps[k + 4 * i + 2], ps[k + 4 * i + 3]))
                                          for i in range(n + 1)])
        else:
            self.filters = nn.ModuleList([_FilterShell(_Holder(ps[k + 2 * i], ps[k + 2 * i + 1]))
                                          for i in range(n + 1)])
        # which parameter tensors are live
        if self._multiscale:
            head_stages = [s for s in (1, 3, 5, 7) if s <= n]
            S = max(head_stages) + 1
            live = [i < S - 1 for i in range(n) for _ in (0, 1)]
            live += [i in head_stages for i in range(n + 1) for _ in (0, 1)]
            live += [i < S for i in range(n + 1) for _ in (0, 1)]
            self.output_layers = head_stages
        else:
            live = [True] * len(ps)
        self._live = live

    def bind_encoder(self, encoder) -> "_MFNBase":
        """Optional: lets the model be called on raw coordinates [B,3] with the gauss encoder fused into the
        filters (what the trainers do).  Without it the model takes the encoded input, like the reference's."""
        if encoder.embedding_type != "gauss":
            raise NotImplementedError("only the 'gauss' Positional_Encoder can be fused into the filters; call the "
                                      "model on encoder.embedding(coords) instead")
        if 2 * encoder.B.shape[0] != self._in:
            raise RuntimeError(f"encoder gives {2 * encoder.B.shape[0]} features, network_input_size is {self._in}")
        self._enc_B = encoder.B.contiguous()
        self._eng = None
        return self

    def _mode_of(self, x: torch.Tensor) -> str:
        """'x': the reference's contract, x = encoder.embedding(coords) [B, network_input_size];
        'gauss': raw coordinates [B,3] with the bound encoder fused."""
        if x.dim() != 2:
            raise RuntimeError(f"model input must be [B, features], got {tuple(x.shape)}")
        if x.shape[1] == self._in:
            return "x"
        if self._enc_B is not None and x.shape[1] == self._enc_B.shape[1]:
            return "gauss"
        raise RuntimeError(f"model input has {x.shape[1]} features: expected the encoded input [B,{self._in}] "
                           "(encoder.embedding(coords), train.py:163-169)"
                           + ("" if self._enc_B is None else f" or raw coordinates [B,{self._enc_B.shape[1]}]"))

    def _engine(self, mode: Optional[str] = None) -> MFNEngine:
        if not self._flat.is_cuda:
            raise RuntimeError("inr_mi355x models run on an MI355X only: call .to('cuda') first (no CPU fallback)")
        if mode is None:
            mode = "gauss" if self._enc_B is not None else "x"
        if mode == "gauss" and self._enc_B is None:
            raise RuntimeError("call bind_encoder(Positional_Encoder(...)) first to run on raw coordinates")
        if self._eng is None:
            self._eng = {}
        if mode not in self._eng:
            from . import _lib as L
            kind = L.KIND_MSBOUNDED if self._bounds is not None else (L.KIND_MSFOURIER if self._multiscale else L.KIND_FOURIER)
            if self._gabor is not None:
                kind = self._kind
            if mode == "gauss":
                eng = MFNEngine(kind, self._in, self._W, self._n, self._out, self._enc_B.shape[0], self._bounds)
            else:
                eng = MFNEngine(kind, self._in, self._W, self._n, self._out, 0, self._bounds, input_mode=L.INPUT_X)
            eng.bind(self._flat)
            self._eng[mode] = eng
        return self._eng[mode]

    def fused_engine(self, enc_size: int) -> MFNEngine:
        return self._engine("gauss")

    def _heads(self, x, dist=None):
        return _MFNFunction.apply(self, x, dist, *self._flat_params)

    def _single(self, x):
        out = self._heads(x)[0]
        return torch.sin(out) if self._output_act else out  # MFNBase.forward, mfn.py:40-41


class FourierNet(_MFNBase):
    """mfn.py:61-94."""

    def __init__(self, params, out_size=1.0, input_scale=2.0, weight_scale=1.0, bias=True, output_act=False):
        super().__init__()
        self._output_act = bool(output_act)
        self._build_mfn(params, input_scale / np.sqrt(params["network_depth"] + 1), weight_scale)

    def forward(self, x, dist_to_center=None):
        return self._single(x)


class GaborNet(_MFNBase):
    """mfn.py:133-162: multiplicative filter network with Gabor filters
    sin(F x + c) * exp(-0.5 * gamma_j * |x - mu_j|^2); mu and gamma are trained."""

    def __init__(self, params, input_scale=2, weight_scale=1.0, alpha=6.0, beta=1.0, bias=True, output_act=False):
        super().__init__()
        self._output_act = bool(output_act)
        from . import _lib as L
        self._kind = L.KIND_KGABOR if isinstance(self, KGaborNet) else L.KIND_GABOR
        n = params["network_depth"]
        self._gabor = (alpha / (n + 1), beta)
        self._build_mfn(params, input_scale / np.sqrt(n + 1), weight_scale)

    def forward(self, x, dist_to_center=None):
        return self._single(x)


class KGaborNet(GaborNet):
    """mfn.py:164-204: ``forward(x, dist_to_center)``; the filters receive dist_to_center but never use it
    (with_dist_filtering stays False), so the arithmetic is GaborNet's."""

    def forward(self, x, dist_to_center=None):
        return self._single(x)


class MultiscaleKFourier(_MFNBase):
    """mfn.py:206-267: heads output_linear[i] for i in [1,3,5,7]; returns a list of [B,out] tensors."""

    _multiscale = True

    def __init__(self, params, weight_scale=1.0, bias=True, output_act=False, centered=True,
                 output_layers=(1, 3, 5, 7), reuse_filters=False):
        super().__init__()
        if tuple(output_layers) != (1, 3, 5, 7):
            raise NotImplementedError("output_layers other than [1,3,5,7]")
        self._build_mfn(params, weight_scale / np.sqrt(params["network_depth"] + 1), weight_scale)

    def forward(self, coords, dist_to_center=None, **kw):  # output_act is ignored here, as in mfn.py:255-267
        h = self._heads(coords)
        return [h[k] for k in range(h.shape[0])]


class MultiscaleBoundedFourier(_MFNBase):
    """mfn.py:288-355: MultiscaleKFourier whose hidden Linears are BoundedLinear(bounds[i]) -- rows of h with
    dist outside [lo, hi] are zeroed before the Linear (the bias still reaches them)."""

    _multiscale = True

    def __init__(self, params, weight_scale=1.0, bias=True, output_act=False, centered=True,
                 output_layers=(1, 3, 5, 7), reuse_filters=False, boundaries=None):
        super().__init__()
        if tuple(output_layers) != (1, 3, 5, 7):
            raise NotImplementedError("output_layers other than [1,3,5,7]")
        self._bounds = [tuple(b) for b in boundaries][:params["network_depth"]]
        self._build_mfn(params, weight_scale / np.sqrt(params["network_depth"] + 1), weight_scale)

    def forward(self, coords, dist_to_center=None):
        h = self._heads(coords, dist_to_center)
        return [h[k] for k in range(h.shape[0])]
