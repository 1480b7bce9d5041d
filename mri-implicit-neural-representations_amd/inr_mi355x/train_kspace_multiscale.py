"""Multi-scale fitting driver: the loop of the reference's src/train_kspace_multiscale.py:161-201 over
the fused MFN kernel (MultiscaleKFourier, 4 heads) with optional data parallelism over coordinates.

Per step (reference lines 164-195):  outs = model(enc(coords), dist);  loss = 0.1*ConsistencyLoss(outs,
dist) + sum_k 0.5*loss_fn(out_k, gt)  (limit_kspace is a no-op, SURVEY A.4 #2: every head sees the full
gt);  Adam;  per-epoch LambdaLR.  Radii of the nested discs come from the k-means ring partition
(inr_mi355x/clustering.py = the reference's clustering.py; train_kspace_multiscale.py:73-84): pass ``radii`` or
leave it None to have them computed from ``config["partition"]`` (no_steps, no_models).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .engine import ConsistencySpec, LossSpec
from .evalchain import psnr, reconstruct
from .mfn import MultiscaleBoundedFourier, MultiscaleKFourier
from .networks import Positional_Encoder
from .train import exchange_and_update, lr_factor, set_default_configs, shard_rows, wants_sharded_update


def create_pairs(values: Sequence[float], multiplication_factor: int):
    """train_kspace_multiscale.py:42-47."""
    pairs = [(values[0], values[i + 1]) for i in range(len(values) - 1)]
    return [(p[0], p[1]) for p in pairs for _ in range(multiplication_factor)]


class MultiscaleTrainer:
    def __init__(self, config: dict, image: torch.Tensor, coords: torch.Tensor, dist: torch.Tensor,
                 radii: Optional[Sequence[float]], shape, device, seed: int = 0, rank: int = 0, world: int = 1,
                 process_group=None, mask: Optional[torch.Tensor] = None, mask_seed: Optional[int] = None):
        config = set_default_configs(dict(config))
        if radii is None:  # train_kspace_multiscale.py:73-84
            from .clustering import partition_and_stats
            C, H, W = int(shape[0]), int(shape[1]), int(shape[2])
            part = config["partition"]
            dev_img = image.to(device).reshape(C, H, W, 2)
            self.mx, radii = partition_and_stats(dev_img, coords.to(device).reshape(C, H, W, 3),
                                                 no_steps=part["no_steps"], no_parts=part["no_models"], stat="max")
            self.mx = torch.cat((self.mx.cpu(), torch.ones(1)))
        self.radii = [float(r) for r in radii]
        self.config = config
        self.device = torch.device(device)
        self.rank, self.world, self.pg = rank, world, process_group
        self.shape = shape
        kinds = {"L2": (L.LOSS_L2_HALF, 1.0), "L1": (L.LOSS_L1_HALF, 1.0), "LSL": (L.LOSS_LOGSPACE, 0.5)}
        if config["loss"] not in kinds:
            # HDR / FFL / tanh crash in the reference's multiscale script
            raise NotImplementedError(f"loss {config['loss']!r} in the multiscale loop")
        kind, self.scale = kinds[config["loss"]]
        opts = config.get("loss_opts", {}) or {}
        self.loss = LossSpec(kind, float(opts.get("hdr_eps", 1e-3)), float(opts.get("hdr_ff_sigma", 2.0)),
                             float(opts.get("hdr_ff_factor", 0.5)))
        torch.manual_seed(seed)
        self.encoder = Positional_Encoder(config["encoder"], device=self.device)  # train_kspace_multiscale.py:90
        if config["model"] == "BoundedFourier":  # train_kspace_multiscale.py:93-95
            self.model = MultiscaleBoundedFourier(config["net"], boundaries=create_pairs(list(radii), 2))
        else:
            self.model = MultiscaleKFourier(config["net"])
        self.model = self.model.to(self.device)
        if config["encoder"]["embedding"] == "gauss":  # fused into every filter: the kernels take raw coordinates
            self.model.bind_encoder(self.encoder)
            self.engine = self.model._engine("gauss")
            self.enc_B = self.encoder.B.contiguous()
        else:  # 'LogF' / 'none': the filters read encoder.embedding(coords) from memory (train_kspace_multiscale.py:169)
            self.engine = self.model._engine("x")
            self.enc_B = None
        self.sharded_update = wants_sharded_update(config, self.engine.n_params, world)
        if self.sharded_update:
            self.engine.enable_sharded_update(rank, world)
        self.pairs = create_pairs(list(radii), 1)
        # undersampling / per-coil batches / TV as in the single-scale loop (models/utils.py:102-123;
        # train_kspace_multiscale.py:173-182)
        self.image_full = image.to(self.device).contiguous()
        from .undersampling import Undersampler, parse_undersampling_argument
        method, uparams = parse_undersampling_argument(config["undersampling"])
        if mask is None and method is not None and method.lower() != "none":
            C, H, W = int(shape[0]), int(shape[1]), int(shape[2])
            masked, _, gm = Undersampler(method, seed=mask_seed).apply(image.reshape(C, H, W, 2).cpu(), uparams)
            image, mask = masked.reshape(-1, 2), gm[:, 0].contiguous()
        self.mask_cpu = mask
        # (sampled rows in front of every row, once: no CPU reduction per step -- see INRTrainer)
        self._mask_cum = None if mask is None else [0] + torch.cumsum(mask.to(torch.int64).flatten(), 0).tolist()
        self.mask = mask.to(torch.uint8).to(self.device).contiguous() if mask is not None else None
        self.per_coil = bool(config["per_coil"])
        self.use_tv = bool(config["use_tv"])  # here TV does not depend on a mask (train_kspace_multiscale.py:173)
        if self.use_tv and not self.per_coil:
            raise ValueError("use_tv needs per_coil batches: tv_loss views the batch as one [H,W,2] coil")
        self.n = coords.shape[0]
        self.coords = coords.to(self.device).contiguous()
        self.image = image.to(self.device).contiguous()
        self.dist_cpu = dist.reshape(-1).contiguous()
        self._dist_np = self.dist_cpu.detach().cpu().numpy()
        self.dist = self.dist_cpu.to(self.device)
        self.bs = int(shape[1] * shape[2]) if self.per_coil else int(config["batch_size"])
        self.steps_per_epoch = math.ceil(self.n / self.bs)
        self.global_step = 0
        self._cons = {}
        if "pretrain" in config:
            self.load_checkpoint(torch.load(config["pretrain"], map_location=self.device))

    def _inputs(self, lo: int, hi: int) -> torch.Tensor:
        return self.coords[lo:hi] if self.enc_B is not None else self.encoder.embedding(self.coords[lo:hi]).contiguous()

    def _cons_spec(self, it: int, lo: int, hi: int) -> ConsistencySpec:
        if it not in self._cons:
            # (numpy, not torch: a torch CPU reduction per step leaves its worker threads spinning, which drove the container
            # into its CPU quota -- 87 ms stalls, profiles/r03_config5_steps.txt; here: a first visit of every batch of epoch 0)
            d = self._dist_np[lo:hi]
            inv = []
            for (blo, bhi) in self.pairs[:-1]:
                n_rows = int(np.count_nonzero((d < blo) | (d > bhi)))
                inv.append(1.0 / (2.0 * n_rows) if n_rows else 0.0)  # mse_loss mean over rows x 2 channels
            self._cons[it] = ConsistencySpec(0.1, self.pairs, inv + [0.0], 2)
        return self._cons[it]

    def _tv_step(self, it: int, lo: int, hi: int, count: int) -> torch.Tensor:
        """Per-coil step with TV on the last head (train_kspace_multiscale.py:164-195 with use_tv): forward (stashing)
        -> multi-head loss gradient -> TV gradient added to the last head's -> backward.  Data parallel by image rows
        with a one-row halo, as INRTrainer._tv_step: the halo row only serves the vertical TV pair -- it is masked out
        of the pointwise terms and moved to dist = 0 (inside every disc) for the consistency term."""
        H, W = int(self.shape[1]), int(self.shape[2])
        y0, y1 = shard_rows(0, H, self.rank, self.world)
        if y1 == y0:
            self.engine.grads.zero_()
            return torch.zeros((), device=self.device)
        ye = min(y1 + 1, H)
        slo, sown, shi = lo + y0 * W, lo + y1 * W, lo + ye * W
        x, d = self._inputs(slo, shi), self.dist[slo:shi]
        outs = self.engine.forward(x, self.enc_B, save=True, dist=d)
        m = torch.ones(shi - slo, dtype=torch.uint8, device=self.device) if self.mask is None else self.mask[slo:shi].clone()
        d_loss = d
        if shi > sown:
            m[sown - slo:] = 0
            d_loss = d.clone()
            d_loss[sown - slo:] = 0
        _, douts = self.engine.loss_grad_multi(self.loss, outs, self.image[slo:shi], count, mask=m, dist=d_loss,
                                               scale=self.scale, cons=self._cons_spec(it, lo, hi))
        loss = self.engine.tv_grad(outs[-1], douts[-1], y1 - y0, W, H)  # adds to the loss word and to douts[-1]
        self.engine.backward(x, self.enc_B, douts, dist=d)
        return loss

    def step(self, epoch: int, it: int) -> torch.Tensor:
        lo, hi = it * self.bs, min((it + 1) * self.bs, self.n)
        count = hi - lo if self._mask_cum is None else self._mask_cum[hi] - self._mask_cum[lo]
        if self.use_tv:
            loss = self._tv_step(it, lo, hi, count)
        else:
            slo, shi = shard_rows(lo, hi, self.rank, self.world)
            if shi == slo:  # a short last batch can leave a rank without rows: it contributes zeros to the sum
                self.engine.grads.zero_()
                loss = torch.zeros((), device=self.device)
            else:
                loss = self.engine.train_step(self._inputs(slo, shi), self.enc_B, self.image[slo:shi], self.loss,
                                              count=count, mask=None if self.mask is None else self.mask[slo:shi],
                                              dist=self.dist[slo:shi], scale=self.scale,
                                              cons=self._cons_spec(it, lo, hi))
        lr = self.config["lr"] * lr_factor(epoch, self.config["max_epoch"])
        loss = exchange_and_update(self.engine, loss, self.world, self.pg, self.sharded_update, lr, self.config["beta1"],
                                   self.config["beta2"], 1e-8, self.config["weight_decay"])
        self.global_step += 1
        return loss

    def checkpoint(self) -> dict:
        """{'net','enc','opt'} as the reference saves it (train_kspace_multiscale.py, same as train.py:247-250)."""
        from .checkpoint import save_dict
        return save_dict(self.model, self.encoder, self.engine, self.config)

    def load_checkpoint(self, ckpt: dict) -> None:
        """config['pretrain'] (train_kspace_multiscale.py:124-128)."""
        from .checkpoint import load_dict

        def rebind(enc):
            if self.enc_B is not None:
                self.enc_B = enc.B.contiguous()
                self.model._enc_B = self.enc_B

        load_dict(self.model, self.encoder, self.engine, ckpt, rebind)

    def fit(self, max_steps: Optional[int] = None, log_every: int = 0):
        logged = []
        for epoch in range(self.config["max_epoch"]):
            for it in range(self.steps_per_epoch):
                if max_steps is not None and self.global_step >= max_steps:
                    return logged
                loss = self.step(epoch, it)
                if log_every and self.global_step % log_every == 0:
                    logged.append((self.global_step, float(loss)))
        return logged

    @torch.no_grad()
    def predict_all(self, chunk: int = 1 << 18) -> torch.Tensor:
        """outs[-1] is the reconstruction (train_kspace_multiscale.py:225)."""
        outs = []
        for lo in range(0, self.n, chunk):
            hi = min(lo + chunk, self.n)
            outs.append(self.engine.forward(self._inputs(lo, hi), self.enc_B, save=False, dist=self.dist[lo:hi])[-1])
        return torch.cat(outs, 0)

    @torch.no_grad()
    def evaluate(self) -> float:
        ref = reconstruct(self.image_full, self.shape, False)
        return float(psnr(ref, reconstruct(self.predict_all(), self.shape, False)))


def main():
    """CLI with the reference's flags (train_kspace_multiscale.py:50-52): --config, --output_path; the scan
    comes from datasets.py, or a synthetic k-space with --synthetic C,H,W."""
    import argparse
    import json
    import os
    import time

    from .synthetic import make_kspace
    from .train import get_config
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--data_samples", type=str, default="")
    ap.add_argument("--output_path", type=str, default=".")
    ap.add_argument("--synthetic", type=str, default=None,
                    help="C,H,W: fit a synthetic k-space of that shape instead of the scan the config names")
    ap.add_argument("--max_steps", type=int, default=None)
    opts = ap.parse_args()
    config = set_default_configs(get_config(opts.config))
    if config["model"] not in ("BoundedFourier",):
        config["model"] = "MultiscaleKFourier"  # train_kspace_multiscale.py:93-98: anything else is the unbounded net
    if opts.synthetic:
        C, H, W = (int(v) for v in opts.synthetic.split(","))
        image, coords, shape = make_kspace(C, H, W, normalization=config.get("normalization", "max"))
    else:  # train_kspace_multiscale.py:57-72: the scan named by config['data_root'/'data'/'set'/'sample'/'slice']
        from .datasets import from_config, trainer_inputs
        image, coords, shape = trainer_inputs(from_config(config, "cuda"))
    dist = torch.sqrt(coords[:, 1] ** 2 + coords[:, 2] ** 2)
    tr = MultiscaleTrainer(config, image, coords, dist, None, shape, "cuda")
    t0 = time.time()
    tr.fit(opts.max_steps, log_every=config.get("log_iter", 20))
    torch.cuda.synchronize()
    res = {"steps": tr.global_step, "seconds": time.time() - t0, "psnr": tr.evaluate(), "radii": tr.radii}
    os.makedirs(opts.output_path, exist_ok=True)
    torch.save(tr.checkpoint(), os.path.join(opts.output_path, "model_%06d.pt" % tr.global_step))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
