"""Single-scale fitting driver: the loop of the reference's src/train.py:155-251 re-expressed
over the fused MI355X engine (tier 2), with optional data parallelism over coordinates.

Kept from the reference, on purpose: sequential unshuffled batches
(batch i = rows [i*bs,(i+1)*bs) of the C-major grid, last batch short), per-EPOCH LambdaLR
``lr*0.2**min(epoch/max_epoch,1)``, Adam with L2-style weight decay, loss on masked rows only
when undersampled (forward still runs on every row), ``psnr`` with max(x).
Changed, on purpose: inputs stay resident in HBM (no per-step H2D, no per-item DataLoader
collate, no ``.item()`` sync per step), the encoder is fused into layer 0, and unsupported
config values raise instead of falling through.

CLI (same flags as the reference, train.py:255-258):
    python -m inr_mi355x.train --config cfg.yaml [--output_path out] [--synthetic C,H,W]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import time
from typing import Optional

import numpy as np
import torch
import yaml

from . import _lib as L
from .engine import LossSpec
from .evalchain import psnr, reconstruct
from .mfn import FourierNet, GaborNet, KGaborNet
from .networks import FFN, SIREN, WIRE, WIRE2D, Positional_Encoder
from .synthetic import make_kspace
from .undersampling import Undersampler, parse_undersampling_argument

MODELS = {"SIREN": SIREN, "FFN": FFN, "WIRE": WIRE, "WIRE2D": WIRE2D,  # train.py:55-68
          "Fourier": FourierNet, "Gabor": GaborNet, "KGabor": KGaborNet}
MFN_MODELS = ("Fourier", "Gabor", "KGabor")


def get_config(path: str) -> dict:
    """models/utils.py:25-32."""
    if not path:
        return {}
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.Loader)


def set_default_configs(config: dict) -> dict:
    """utils.py:7-23."""
    config.setdefault("per_coil", False)
    config.setdefault("use_tv", False)
    if "regularization" not in config:
        config["regularization"] = {"type": "none"}
    config.setdefault("undersampling", None)
    return config


def lr_factor(epoch: int, max_epoch: int) -> float:
    """LambdaLR lambda of train.py:153."""
    return 0.2 ** min(epoch / max_epoch, 1)


def shard_rows(lo: int, hi: int, rank: int, world: int):
    """Contiguous split of batch rows [lo,hi) over ranks: rank r gets
    [lo + r*n//world, lo + (r+1)*n//world)."""
    n = hi - lo
    return lo + (rank * n) // world, lo + ((rank + 1) * n) // world


def allreduce_step_outputs(grads: torch.Tensor, loss: torch.Tensor, world: int, group=None,
                           gbuf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The one exchange step of the data-parallel path: every rank holds the partial
    sums of its contiguous row shard, already divided by the GLOBAL row count, so a plain SUM
    all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) of the flat fp32 gradient and
    of the loss scalar reproduces the single-GPU step up to summation order.  In place on `grads`.
    ``gbuf`` = the engine's [P+1] buffer whose first P words ARE ``grads``: the loss rides in the last word and
    the step costs one collective instead of two (the message is small, so latency is what counts)."""
    if world <= 1:
        return loss
    import torch.distributed as dist
    if gbuf is not None:
        assert gbuf.data_ptr() == grads.data_ptr() and gbuf.numel() == grads.numel() + 1
        if loss.data_ptr() != gbuf[-1:].data_ptr():  # the fused steps already left their loss in that word
            gbuf[-1:].copy_(loss.reshape(1))
        dist.all_reduce(gbuf, op=dist.ReduceOp.SUM, group=group)
        return gbuf[-1].clone()
    dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    loss = loss.clone()
    dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
    return loss


# Below this the replicated update is cheaper than the sharded one's two collectives, chunk copies and re-pack launch:
# measured on one MI355X, Adam + re-pack of BASELINE config 4's 4.47 M parameters is 0.038 ms replicated against 0.039 ms of
# local work per rank sharded 8 ways -- the sharded update starts to pay at several times that size.
SHARDED_UPDATE_MIN_PARAMS = 1 << 24


def wants_sharded_update(config: dict, n_params: int, world: int) -> bool:
    """config["dp_sharded_update"] (True / False; default: networks of >= 2^24 parameters -- none of the BASELINE
    configs) decides between the two exchange steps of exchange_and_update."""
    v = config.get("dp_sharded_update")
    if v is None:
        return world > 1 and n_params >= SHARDED_UPDATE_MIN_PARAMS
    if world <= 1:  # an explicit True on one rank runs the same collectives over a one-rank group (rehearsals of the nccl calls)
        import torch.distributed as dist
        return bool(v) and dist.is_available() and dist.is_initialized()
    return bool(v)


def exchange_and_update(engine, loss: torch.Tensor, world: int, group, sharded: bool, lr: float, beta1: float,
                        beta2: float, eps: float = 1e-8, weight_decay: float = 0.0, l1: float = 0.0,
                        l2: float = 0.0, cplx_reg: Optional[tuple] = None) -> torch.Tensor:
    """Exchange step + optimizer step of one data-parallel iteration (single process in the reference:
    train.py:189-190, train_kspace_multiscale.py:199-200).  Replicated: ONE all-reduce of [gradient | loss], then every
    rank runs the whole Adam update.  Sharded: reduce-scatter, Adam on 1/N of the entries, all-gather, re-pack
    (MLPEngine.adam_step_sharded) -- the same bytes on the links, 1/N of the update per rank.  ``cplx_reg`` = (l1, l2, l2_dir) replaces l1 / l2
    for models with complex64 tensors (INRTrainer._penalty).  Returns the global loss."""
    if sharded:
        if loss.data_ptr() != engine._loss_word.data_ptr():
            engine._loss_word.copy_(loss.reshape(1))
        if cplx_reg is not None:
            return engine.adam_step_sharded(group, lr, beta1, beta2, eps, weight_decay, reg=cplx_reg)
        return engine.adam_step_sharded(group, lr, beta1, beta2, eps, weight_decay, l1, l2)
    loss = allreduce_step_outputs(engine.grads, loss, world, group, engine.gbuf)
    if cplx_reg is not None:  # (l1, l2, l2_dir) of a model with complex tensors: the penalty gradient by inr_reg_grad, once,
        engine.reg_grad(*cplx_reg)  # on the summed gradient (every rank holds the same parameters)
        engine.adam_step(lr, beta1, beta2, eps, weight_decay)
    else:
        engine.adam_step(lr, beta1, beta2, eps, weight_decay, l1, l2)
    return loss


def center_pair_rows(kcoords: torch.Tensor, min_sample: int, n_bands: int = 2):
    """The row pairs of CenterLoss's N_BANDS radial bands (losses.py:176-194), as the reference draws them: band k
    compares dist^2 = ky^2 + kx^2 with the RATIOS (k-1)/N (0.1 for the first band) and k/N; n = min(min_sample, |inner|,
    |ring|); torch.randperm on the default CPU generator, inner set first.  Yields (rows_a, rows_b) int64 on kcoords'
    device; bands with an empty side are skipped."""
    d2 = kcoords[:, 1] ** 2 + kcoords[:, 2] ** 2
    for band in range(1, n_bands + 1):
        r1 = (band - 1) / n_bands or 0.1
        m1 = d2 <= r1
        m2 = (d2 <= band / n_bands) & ~m1
        rows1, rows2 = torch.nonzero(m1)[:, 0], torch.nonzero(m2)[:, 0]
        n = min(min_sample, rows1.numel(), rows2.numel())
        if n == 0:
            continue
        a = torch.randperm(rows1.numel())[:n].to(kcoords.device)
        b = torch.randperm(rows2.numel())[:n].to(kcoords.device)
        yield rows1[a].contiguous(), rows2[b].contiguous()


class INRTrainer:
    def __init__(self, config: dict, image: torch.Tensor, coords: torch.Tensor, shape, device,
                 seed: int = 0, mask: Optional[torch.Tensor] = None, rank: int = 0, world: int = 1,
                 process_group=None, mask_seed: Optional[int] = None, graph_steps: bool = False):
        config = set_default_configs(dict(config))
        self.config = config
        self.device = torch.device(device)
        self.rank, self.world, self.pg = rank, world, process_group
        self.shape = shape
        if config["model"] not in MODELS:
            raise NotImplementedError(f"model {config['model']!r} has no MI355X kernel yet (have {sorted(MODELS)})")
        if config.get("optimizer", "Adam") != "Adam":
            raise NotImplementedError("only Adam (train.py:75-78)")
        # construction order and RNG use of train.py:52-71: encoder, then model, on the CPU generator
        torch.manual_seed(seed)
        self.encoder = Positional_Encoder(config["encoder"], device=self.device)
        self.model = MODELS[config["model"]](config["net"]).to(self.device)
        emb = config["encoder"]["embedding"]
        self.is_mfn = config["model"] in MFN_MODELS
        if self.is_mfn and getattr(self.model, "_output_act", False):
            raise NotImplementedError("output_act in the fused training step (no shipped config sets it)")
        if self.is_mfn and emb == "gauss":  # fused into every filter: the model runs on raw coordinates
            self.model.bind_encoder(self.encoder)
            self.engine = self.model._engine("gauss")
            self.enc_B = self.encoder.B.contiguous()
        elif self.is_mfn:  # 'LogF' / 'none': the filters read encoder.embedding(coords) from memory
            self.engine = self.model._engine("x")
            self.enc_B = None
        elif emb == "gauss":  # config["precision"]: "f32" (parity path, default) | "bf16" (throughput path)
            self.engine = self.model.fused_engine(config["encoder"]["embedding_size"],
                                                  **({"precision": config["precision"]} if "precision" in config else {}))
            self.enc_B = self.encoder.B.contiguous()
        else:
            self.engine = self.model._engine()
            self.enc_B = None
        self.sharded_update = wants_sharded_update(config, self.engine.n_params, world)
        if self.sharded_update:
            self.engine.enable_sharded_update(rank, world)
        self.loss = LossSpec.from_config(config)
        reg = config["regularization"]
        self.l1 = float(reg["strenght"]) if reg["type"] == "L1" else 0.0
        self.l2 = float(reg["strenght"]) if reg["type"] == "L2" else 0.0
        if reg["type"] not in ("none", "L1", "L2"):
            raise NotImplementedError(f"regularization {reg['type']!r}")
        # regularization.py:21-36 on complex64 tensors means sum |z| (L1) and |sum z^2| (L2, a complex square), and
        # model.parameters() includes the frozen omega_0 / scale_0 (networks.py:191-192): the Adam kernel's per-entry sign /
        # 2p terms are the real-parameter forms, so these models take the penalty gradient from inr_reg_grad (_penalty)
        self._cplx_reg = bool(self.l1 or self.l2) and any(c for (_, _, _, c) in self.model._layout)
        if self._cplx_reg:
            P = self.engine.n_params
            sign = torch.ones(P)
            re_idx = []
            for (o, n, _, c) in self.model._layout:
                if c:
                    sign[o + 1:o + n:2] = -1.0
                    re_idx.append(torch.arange(o, o + n, 2))
            self._sq_sign = sign.to(self.device)  # p^2 enters Re(S) with +1 (real entries, real parts) or -1 (imaginary parts)
            self._re_idx = torch.cat(re_idx).to(self.device)
            is_real = torch.ones(P, dtype=torch.bool)
            for (o, n, _, c) in self.model._layout:
                if c:
                    is_real[o:o + n] = False
            self._real_idx = torch.nonzero(is_real)[:, 0].to(self.device)
            frozen = [p.detach().double().cpu() for p in self.model.parameters()
                      if not any(p is q for q in self.model._flat_params)]
            self._frozen_l1 = float(sum(f.abs().sum() for f in frozen))
            self._frozen_l2 = float(sum((f * f).sum() for f in frozen))
        # undersampled fit (models/utils.py:102-123): train on the zero-filled k-space with the loss on
        # sampled rows only; validation still compares with the full k-space (val_loader, utils.py:131-137)
        self.image_full = image.to(self.device).contiguous()
        method, uparams = parse_undersampling_argument(config["undersampling"])
        if mask is None and method is not None and method.lower() != "none":
            C, H, W = shape[0], shape[1], shape[2]
            us = Undersampler(method, seed=mask_seed)
            masked, _, gm = us.apply(image.reshape(C, H, W, 2).cpu(), uparams)
            image, mask = masked.reshape(-1, 2), gm[:, 0].contiguous()
        # data resident in HBM for the whole fit
        self.n = coords.shape[0]
        self.coords = coords.to(self.device).contiguous()
        self.image = image.to(self.device).contiguous()
        self.mask_cpu = mask
        # sampled rows in front of every row, once: a batch's count is a difference of two entries.  (A `mask[lo:hi].sum()`
        # per step is a multi-threaded CPU reduction whose worker threads spin on after it: on the GPU boxes that drove the
        # container into its CPU quota -- 87 ms stalls every ~17 steps of the per-coil loop, profiles/r03_config5_steps.txt)
        self._mask_cum = None
        if mask is not None:
            # (a numpy int64 array, not a Python list: 15 coils are 3.5 M entries -- a list of ints of that length is > 100 MB)
            cum = np.zeros(mask.numel() + 1, dtype=np.int64)
            np.cumsum(mask.flatten().to(torch.int64).numpy(), out=cum[1:])
            self._mask_cum = cum
        self.mask = mask.to(torch.uint8).to(self.device).contiguous() if mask is not None else None
        # per-coil batches (MRICoilWrapperDataset, nerp_datasets.py:397-441; loader batch_size 1 = one coil,
        # models/utils.py:65-66) so that TV can see a whole coil grid
        self.per_coil = bool(config["per_coil"])
        self.bs = int(shape[1] * shape[2]) if self.per_coil else int(config["batch_size"])
        self.use_tv = bool(config["use_tv"]) and self.mask is not None  # train.py:172-175: only inside the mask branch
        if self.use_tv and not self.per_coil:
            raise ValueError("use_tv needs per_coil batches: tv_loss views the batch as one [H,W,2] coil (train.py:175)")
        if self.loss.kind == L.LOSS_CENTER:
            if self.mask is not None:
                # the reference indexes the MASKED predictions with radial masks of the UNMASKED coordinates
                # (train.py:176-179, losses.py:188-189): an IndexError there
                raise NotImplementedError("loss 'LSL' (CenterLoss) with an undersampling mask")
            if self.is_mfn or world > 1 or self.per_coil:
                raise NotImplementedError("loss 'LSL' (CenterLoss): single-rank SIREN / FFN / WIRE fits on plain batches")
        self.steps_per_epoch = math.ceil(self.n / self.bs)
        self.global_step = 0
        self._hdr_A = {}
        # graph_steps: every batch of the epoch becomes one captured HIP graph (fused kernel, weight-gradient GEMM,
        # reduction, Adam, step advance) replayed from then on -- the batches are fixed views of the resident data
        # (sequential sampler, models/utils.py:126-130), the step count and learning rate live in device memory.
        # Single rank only (the gradient all-reduce sits between the two halves) and only where a step is ONE
        # fused launch sequence on resident views.
        self.graph_steps = bool(graph_steps) and world == 1 and not self.use_tv and \
            self.loss.kind != L.LOSS_CENTER and (self.enc_B is not None or emb == "none") and not self._cplx_reg
        self._graphs = {}
        # plain single-rank steps of the MLP engines go through inr_train_adam_step (INR_ONE_CALL_STEPS=0: two calls)
        self.one_call_steps = (os.environ.get("INR_ONE_CALL_STEPS", "1") != "0" and not self.is_mfn and not self.use_tv
                               and self.loss.kind != L.LOSS_CENTER and not self._cplx_reg)
        if "pretrain" in config:  # train.py:117-121
            self.load_checkpoint(torch.load(config["pretrain"], map_location=self.device))

    # ---- one optimizer step on batch `it` of epoch `epoch` --------------------------------------
    def _inputs(self, lo: int, hi: int):
        if self.enc_B is not None:
            return self.coords[lo:hi]
        if self.config["encoder"]["embedding"] == "none":
            return self.coords[lo:hi]
        return self.encoder.embedding(self.coords[lo:hi])

    def _batch_hdr_A(self, it: int, lo: int, hi: int) -> float:
        """A = mean_i((1-f_i)^2) over ALL batch coordinates (losses.py:241-242,258)."""
        if self.loss.kind not in (L.LOSS_HDR, L.LOSS_CENTER):
            return 0.0
        if it not in self._hdr_A:
            kc = self.coords[lo:hi]
            f = torch.exp(-(kc[:, 1] ** 2 + kc[:, 2] ** 2) / (2 * self.loss.sigma ** 2))
            self._hdr_A[it] = float(torch.mean((1 - f) ** 2))
        return self._hdr_A[it]

    def _count(self, lo: int, hi: int) -> int:
        """sampled rows of [lo, hi) (all of them without a mask)"""
        return hi - lo if self._mask_cum is None else int(self._mask_cum[hi] - self._mask_cum[lo])

    def _penalty(self):
        """(value, cplx_reg): the penalty VALUE the reference adds to the logged loss, at the parameters the step starts from
        (train.py:185-192) -- None without a regulariser -- and, for models with complex64 tensors, the (l1, l2, l2_dir)
        that exchange_and_update hands to inr_reg_grad: regularization.py:25-28 is sum |z| over complex entries,
        :34-36 is |S| with S = sum p^2 a complex number (z^2 = a^2 - b^2 + 2iab), over EVERY Parameter, the frozen omega_0 /
        scale_0 included; l2_dir = conj(S) / |S| stays on the device."""
        if not (self.l1 or self.l2):
            return None, None
        p = self.engine.params
        if not self._cplx_reg:
            return (self.l1 * p.abs().sum() if self.l1 else self.l2 * (p * p).sum()), None
        a, b = p[self._re_idx], p[self._re_idx + 1]
        if self.l1:
            return self.l1 * (p[self._real_idx].abs().sum() + torch.hypot(a, b).sum() + self._frozen_l1), (self.l1, 0.0, None)
        s_re = (self._sq_sign * p * p).sum() + self._frozen_l2
        s_im = 2.0 * (a * b).sum()
        mod = torch.hypot(s_re, s_im)
        return self.l2 * mod, (0.0, self.l2, (torch.stack((s_re, -s_im)) / mod).contiguous())

    def step(self, epoch: int, it: int) -> torch.Tensor:
        lo, hi = it * self.bs, min((it + 1) * self.bs, self.n)
        count = self._count(lo, hi)
        A = self._batch_hdr_A(it, lo, hi)
        if self.graph_steps:
            return self._graph_step(epoch, it, lo, hi, count, A)
        if self.world == 1 and self.one_call_steps and not self.sharded_update and hi > lo:
            # single rank: nothing sits between the reduction and the update -- one call, one launch less
            cfg = self.config
            penalty, _ = self._penalty()  # value of the penalty at the parameters the step starts from, as below
            m = self.mask[lo:hi] if self.mask is not None else None
            loss = self.engine.train_adam_step(self._inputs(lo, hi), self.enc_B, self.image[lo:hi], self.loss,
                                               cfg["lr"] * lr_factor(epoch, cfg["max_epoch"]), count=count, mask=m,
                                               hdr_A=A, beta1=cfg["beta1"], beta2=cfg["beta2"], eps=1e-8,
                                               weight_decay=cfg["weight_decay"], l1=self.l1, l2=self.l2)
            self.global_step += 1
            return loss if penalty is None else loss + penalty
        if self.use_tv:
            loss = self._tv_step(lo, count, A)
        elif self.loss.kind == L.LOSS_CENTER:
            loss = self._center_step(lo, hi, A)
        else:
            slo, shi = shard_rows(lo, hi, self.rank, self.world)
            if shi == slo:  # a short last batch can leave a rank without rows: it contributes zeros to the sum
                self.engine.grads.zero_()
                loss = torch.zeros((), device=self.device)
            else:
                m = self.mask[slo:shi] if self.mask is not None else None
                loss = self._fused(slo, shi, count, m, A)
        # the loss the reference logs includes the penalty VALUE at the parameters the step starts from
        # (train.py:185-192); its gradient is formed inside the Adam kernel.  Every rank holds the same parameters.
        penalty, cplx_reg = self._penalty()
        lr = self.config["lr"] * lr_factor(epoch, self.config["max_epoch"])
        loss = exchange_and_update(self.engine, loss, self.world, self.pg, self.sharded_update, lr, self.config["beta1"],
                                   self.config["beta2"], 1e-8, self.config["weight_decay"], self.l1, self.l2, cplx_reg)
        self.global_step += 1
        return loss if penalty is None else loss + penalty

    def _graph_step(self, epoch: int, it: int, lo: int, hi: int, count: int, A: float) -> torch.Tensor:
        cfg = self.config
        lr = cfg["lr"] * lr_factor(epoch, cfg["max_epoch"])
        g = self._graphs.get(it)
        if g is None or g.stale:
            m = self.mask[lo:hi] if self.mask is not None else None
            g = self.engine.capture_step(lambda: self._fused(lo, hi, count, m, A), lr, cfg["beta1"], cfg["beta2"], 1e-8,
                                         cfg["weight_decay"], self.l1, self.l2)
            self._graphs[it] = g
        penalty, _ = self._penalty()  # value of the penalty at the parameters the step starts from, as in step()
        loss = g.replay(lr)
        self.global_step += 1
        return loss if penalty is None else loss + penalty

    def prepare_graphs(self, epoch: int = 0) -> None:
        """Capture every batch of an epoch up front (largest first), so that no capture falls into a timed region.
        Runs each batch's gradient launch once; parameters and the step count do not move."""
        if not self.graph_steps:
            return
        cfg = self.config
        lr = cfg["lr"] * lr_factor(epoch, cfg["max_epoch"])
        for it in range(self.steps_per_epoch):
            lo, hi = it * self.bs, min((it + 1) * self.bs, self.n)
            count = self._count(lo, hi)
            A = self._batch_hdr_A(it, lo, hi)
            g = self._graphs.get(it)
            if g is None or g.stale:
                m = self.mask[lo:hi] if self.mask is not None else None
                self._graphs[it] = self.engine.capture_step(
                    lambda lo=lo, hi=hi, count=count, m=m, A=A: self._fused(lo, hi, count, m, A), lr, cfg["beta1"],
                    cfg["beta2"], 1e-8, cfg["weight_decay"], self.l1, self.l2)

    def _fused(self, slo, shi, count, m, A):
        return self.engine.train_step(self._inputs(slo, shi), self.enc_B, self.image[slo:shi], self.loss,
                                      count=count, mask=m, hdr_A=A)

    def _center_step(self, lo: int, hi: int, A: float) -> torch.Tensor:
        """CenterLoss ('LSL', train.py:87-88,178-180; losses.py:141-201): forward -> pointwise part -> the random-pair
        term of the two radial bands -> backward.  The pairs are drawn with torch.randperm on the CPU generator in the
        reference's order (band 1: inner, ring; band 2: inner, ring), from masks on dist^2 = ky^2 + kx^2 compared with the
        band RATIOS (losses.py:153-154,180-187).  Single rank, whole batches (pairs span the batch)."""
        x, gt = self._inputs(lo, hi), self.image[lo:hi]
        out = self.engine.forward(x, self.enc_B, save=True)
        loss, dout = self.engine.loss_grad(self.loss, out, gt, hi - lo, hdr_A=A)
        for rows_a, rows_b in center_pair_rows(self.coords[lo:hi], self.loss.min_sample):
            loss = self.engine.center_pairs_grad(out, gt, dout, rows_a, rows_b, 0.1)
        self.engine.backward(x, self.enc_B, dout)
        return loss

    def _tv_step(self, lo: int, count: int, A: float) -> torch.Tensor:
        """Per-coil step with total variation (train.py:163-189 with use_tv): forward (stashing) ->
        masked pointwise loss -> TV added on the whole coil grid -> backward.  Data parallel: image rows
        are split over ranks; each rank also evaluates one halo row below its slab so that every vertical
        TV pair is owned by exactly one rank (the halo row's pointwise loss stays with its owner)."""
        H, W = int(self.shape[1]), int(self.shape[2])
        y0, y1 = shard_rows(0, H, self.rank, self.world)
        if y1 == y0:  # more ranks than image rows: nothing to contribute
            self.engine.grads.zero_()
            return torch.zeros((), device=self.device)
        ye = min(y1 + 1, H)
        slo, shi = lo + y0 * W, lo + ye * W
        out = self.engine.forward(self._inputs(slo, shi), self.enc_B, save=True)
        if self.is_mfn:  # the filter networks' engine hands back [heads = 1, B, out] (train.py:165-169 calls model(coords))
            out = out[0]
        # pointwise loss on the owned rows' sampled coordinates + TV on the grid, one pass (inr_loss_tv_grad)
        loss, dout = self.engine.loss_tv_grad(self.loss, out, self.image[slo:shi], count, y1 - y0, W, H,
                                              mask=self.mask[slo:shi], hdr_A=A)
        self.engine.backward(self._inputs(slo, shi), self.enc_B, dout.unsqueeze(0) if self.is_mfn else dout)
        return loss

    def fit(self, max_steps: Optional[int] = None, log_every: int = 0):
        """Runs epochs of sequential batches (train.py:155-198).  Returns the list of losses logged."""
        logged = []
        for epoch in range(self.config["max_epoch"]):
            for it in range(self.steps_per_epoch):
                if max_steps is not None and self.global_step >= max_steps:
                    return logged
                loss = self.step(epoch, it)
                if log_every and self.global_step % log_every == 0:
                    logged.append((self.global_step, float(loss)))
        return logged

    # ---- validation (train.py:199-231) -----------------------------------------------------------
    @torch.no_grad()
    def predict_all(self, chunk: int = 1 << 18) -> torch.Tensor:
        outs = []
        for lo in range(0, self.n, chunk):
            hi = min(lo + chunk, self.n)
            o = self.engine.forward(self._inputs(lo, hi), self.enc_B, save=False)
            outs.append(o[0] if self.is_mfn else o)
        return torch.cat(outs, 0)

    @torch.no_grad()
    def evaluate(self) -> float:
        in_image_space = bool(self.config.get("transform", False))
        ref = reconstruct(self.image_full, self.shape, in_image_space)
        rec = reconstruct(self.predict_all(), self.shape, in_image_space)
        return float(psnr(ref, rec))

    def checkpoint(self) -> dict:
        """Same dict as train.py:247-250 ('opt' in torch.optim.Adam.state_dict() layout)."""
        from .checkpoint import save_dict
        return save_dict(self.model, self.encoder, self.engine, self.config)

    def load_checkpoint(self, ckpt: dict) -> None:
        """train.py:117-121 (config['pretrain']): weights, Adam moments / step count and the encoder matrix."""
        from .checkpoint import load_dict

        def rebind(enc):
            if self.enc_B is not None:
                self.enc_B = enc.B.contiguous()
            if self.is_mfn:
                self.model._enc_B = self.enc_B  # the engine (and its Adam state) stays; B is passed per call

        load_dict(self.model, self.encoder, self.engine, ckpt, rebind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--data_samples", type=str, default="")
    ap.add_argument("--output_path", type=str, default=".")
    ap.add_argument("--synthetic", type=str, default=None,
                    help="C,H,W: fit a synthetic k-space of that shape instead of the scan the config names")
    ap.add_argument("--max_steps", type=int, default=None)
    opts = ap.parse_args()
    config = set_default_configs(get_config(opts.config))
    if opts.synthetic:
        C, H, W = (int(v) for v in opts.synthetic.split(","))
        image, coords, shape = make_kspace(C, H, W, normalization=config.get("normalization", "coil"),
                                           image_space=bool(config.get("transform", False)))
    else:  # train.py:271-287: config['data_root'/'data'/'set'/'sample'/'slice'] (or 'custom_file_or_path')
        from .datasets import from_config, trainer_inputs
        image, coords, shape = trainer_inputs(from_config(config, "cuda"))
    tr = INRTrainer(config, image, coords, shape, "cuda")
    t0 = time.time()
    tr.fit(opts.max_steps, log_every=config.get("log_iter", 20))
    torch.cuda.synchronize()
    res = {"steps": tr.global_step, "seconds": time.time() - t0, "psnr": tr.evaluate()}
    os.makedirs(opts.output_path, exist_ok=True)
    torch.save(tr.checkpoint(), os.path.join(opts.output_path, "model_%06d.pt" % tr.global_step))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
