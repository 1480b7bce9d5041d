"""Checkpoint interchange with the reference: the dict ``{'net', 'enc', 'opt'}`` of train.py:247-250 /
train_kspace_multiscale.py, with ``'opt'`` in ``torch.optim.Adam.state_dict()`` layout (state indexed by the
position of the parameter in ``model.parameters()``; frozen / never-stepped parameters have no entry), and the
``config["pretrain"]`` resume path of train.py:117-121.
"""
from typing import Optional

import torch

from .networks import _view


def _param_positions(model):
    """position in model.parameters() of every flat-buffer parameter, in layout order."""
    pos = {id(p): j for j, p in enumerate(model.parameters())}
    return [pos[id(p)] for p in model._flat_params]


def optimizer_state(model, engine, config: dict) -> dict:
    live = getattr(model, "_live", [True] * len(model._layout))
    state = {}
    for (off, n, shp, cplx), j, lv in zip(model._layout, _param_positions(model), live):
        if engine.step == 0 or not lv:
            continue  # Adam creates state lazily at the first step with a gradient
        state[j] = {"step": torch.tensor(float(engine.step)),
                    "exp_avg": _view(engine.exp_avg, off, n, shp, cplx).clone(),
                    "exp_avg_sq": _view(engine.exp_avg_sq, off, n, shp, cplx).clone()}
    n_params = len(list(model.parameters()))
    group = {"lr": config["lr"], "betas": (config["beta1"], config["beta2"]), "eps": 1e-8,
             "weight_decay": config["weight_decay"], "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "params": list(range(n_params))}
    return {"state": state, "param_groups": [group]}


def save_dict(model, encoder, engine, config: dict) -> dict:
    # clones: the live state_dict entries are views of ONE flat buffer (float and complex views of the same storage
    # cannot be pickled together, and a checkpoint must not alias the running weights)
    net = type(model.state_dict())((k, v.detach().clone()) for k, v in model.state_dict().items())
    return {"net": net, "enc": None if encoder.B is None else encoder.B.clone(),
            "opt": optimizer_state(model, engine, config)}


@torch.no_grad()
def load_dict(model, encoder, engine, ckpt: dict, rebind=None) -> None:
    """model.load_state_dict(ckpt['net']); optim.load_state_dict(ckpt['opt']); encoder.B = ckpt['enc']."""
    model.load_state_dict(ckpt["net"])
    if ckpt.get("enc") is not None and encoder is not None:
        encoder.B = ckpt["enc"].to(encoder.B.device if encoder.B is not None else ckpt["enc"].device)
        if rebind is not None:
            rebind(encoder)  # the fused kernels hold their own contiguous copy of B
    opt: Optional[dict] = ckpt.get("opt")
    engine.exp_avg.zero_()
    engine.exp_avg_sq.zero_()
    engine.step = 0
    if opt:
        steps = set()
        for (off, n, shp, cplx), j in zip(model._layout, _param_positions(model)):
            st = opt["state"].get(j)
            if st is None:
                continue
            _view(engine.exp_avg, off, n, shp, cplx).copy_(st["exp_avg"].to(engine.exp_avg.device))
            _view(engine.exp_avg_sq, off, n, shp, cplx).copy_(st["exp_avg_sq"].to(engine.exp_avg.device))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"parameters with different Adam step counts {sorted(steps)}: the engine keeps one")
        engine.step = steps.pop() if steps else 0
    engine.pack()
