"""Per-ring model ensemble: one network per k-means ring of k-space (the reference's
train_variations/train_clustering.py:121-190 over clustering.partition_kspace), mapped ONE RING PER GPU.

Ring i = coordinates with radii[i] <= dist <= radii[i+1] (both ends included, so boundary points belong to two
rings; at evaluation the outer ring's prediction wins, as the sequential ``batch_rec[ind] = output`` of the
reference, :199-211).  Every model sees every batch and trains on the rows of its ring: forward on all rows, loss
on the ring's rows, mean over the ring's rows (train_clustering.py:170-183) -- which is exactly the fused step
with a row mask.

Parallel mapping: ring i is owned by rank i % world.  Models are independent, so there is NO collective in
training; the evaluation sweep adds the ranks' disjoint contributions with one SUM all-reduce.

Kept different from the stale reference script, on purpose: batches are the sequential unshuffled ranges of the
maintained loops (the script shuffles), and the +-N(0, 0.05) jitter of the ring bounds (:166-167, an unseeded numpy
draw per model per step) is off by default and seeded when enabled.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from .engine import LossSpec
from .evalchain import psnr, reconstruct
from .networks import Positional_Encoder
from .train import MODELS, MFN_MODELS, lr_factor, set_default_configs


def ring_owner(i: int, world: int) -> int:
    return i % world


def winning_ring(dist: torch.Tensor, radii: Sequence[float]) -> torch.Tensor:
    """Index of the LAST ring containing each point (-1: none), i.e. the one whose output survives the
    sequential overwrite of train_clustering.py:199-211."""
    win = torch.full(dist.shape, -1, dtype=torch.int64, device=dist.device)
    for i in range(len(radii) - 1):
        win[(dist >= radii[i]) & (dist <= radii[i + 1])] = i
    return win


class RingEnsembleTrainer:
    def __init__(self, config: dict, image: torch.Tensor, coords: torch.Tensor, shape, device,
                 radii: Optional[Sequence[float]] = None, seed: int = 0, rank: int = 0, world: int = 1,
                 process_group=None, jitter: float = 0.0):
        config = set_default_configs(dict(config))
        self.config, self.device = config, torch.device(device)
        self.rank, self.world, self.pg = rank, world, process_group
        self.shape = shape
        if config["model"] not in MODELS or config["model"] in MFN_MODELS:
            raise NotImplementedError(f"ring ensembles are built from SIREN / FFN / WIRE / WIRE2D, not {config['model']!r}")
        C, H, W = int(shape[0]), int(shape[1]), int(shape[2])
        if radii is None:  # train_clustering.py:121-126
            from .clustering import partition_kspace
            part = config["partition"]
            _, radii = partition_kspace(image.to(self.device).reshape(C, H, W, 2),
                                        coords.to(self.device).reshape(C, H, W, 3),
                                        no_steps=part["no_steps"], no_parts=part["no_models"])
        self.radii = [float(r) for r in radii]
        self.no_models = len(self.radii) - 1
        self.owned = [i for i in range(self.no_models) if ring_owner(i, world) == rank]
        # one shared encoder, then the models in ring order: every rank builds ALL of them so that the RNG stream
        # (and therefore each ring's initial weights) does not depend on the world size; only owned ones move to HBM
        torch.manual_seed(seed)
        self.encoder = Positional_Encoder(config["encoder"], device=self.device)
        emb = config["encoder"]["embedding"]
        self.enc_B = self.encoder.B.contiguous() if emb == "gauss" else None
        self.models, self.engines = {}, {}
        for i in range(self.no_models):
            m = MODELS[config["model"]](config["net"])
            if i in self.owned:
                m = m.to(self.device)
                self.models[i] = m
                self.engines[i] = (m.fused_engine(config["encoder"]["embedding_size"]) if emb == "gauss"
                                   else m._engine())
        self.loss = LossSpec.from_config(config)
        self.n = coords.shape[0]
        self.coords = coords.to(self.device).contiguous()
        self.image = image.to(self.device).contiguous()
        self.dist = torch.sqrt(self.coords[:, 1] ** 2 + self.coords[:, 2] ** 2)
        self.bs = int(config["batch_size"])
        self.steps_per_epoch = math.ceil(self.n / self.bs)
        self.global_step = 0
        self.jitter = float(jitter)
        self._rng = np.random.RandomState(seed)
        self._masks = {}

    def _inputs(self, lo: int, hi: int):
        emb = self.config["encoder"]["embedding"]
        if self.enc_B is not None or emb == "none":
            return self.coords[lo:hi]
        return self.encoder.embedding(self.coords[lo:hi])

    def _ring_mask(self, i: int, lo: int, hi: int):
        r0, r1 = self.radii[i], self.radii[i + 1]
        if self.jitter > 0.0:  # train_clustering.py:166-167 (every rank draws for every ring: same stream everywhere)
            r0 = max(0.0, r0 - abs(self._rng.normal(0, self.jitter)))
            r1 = r1 + abs(self._rng.normal(0, self.jitter))
            d = self.dist[lo:hi]
            m = ((d >= r0) & (d <= r1)).to(torch.uint8)
            return m, int(m.sum())
        key = (i, lo)
        if key not in self._masks:
            d = self.dist[lo:hi]
            m = ((d >= r0) & (d <= r1)).to(torch.uint8).contiguous()
            self._masks[key] = (m, int(m.sum()))
        return self._masks[key]

    def step(self, epoch: int, it: int) -> List[Optional[float]]:
        """One batch through every owned ring model; returns the per-ring losses (None: ring absent from the batch)."""
        lo, hi = it * self.bs, min((it + 1) * self.bs, self.n)
        lr = self.config["lr"] * lr_factor(epoch, self.config["max_epoch"])
        x, gt = self._inputs(lo, hi), self.image[lo:hi]
        out: List[Optional[torch.Tensor]] = [None] * self.no_models
        for i in range(self.no_models):
            mask, cnt = self._ring_mask(i, lo, hi)  # drawn for every ring to keep the jitter stream rank-independent
            if i not in self.engines or cnt == 0:  # train_clustering.py:169: empty ring -> optimizer has nothing to do
                continue
            eng = self.engines[i]
            out[i] = eng.train_step(x, self.enc_B, gt, self.loss, count=cnt, mask=mask).clone()
            eng.adam_step(lr, self.config["beta1"], self.config["beta2"], 1e-8, self.config["weight_decay"])
        self.global_step += 1
        return out

    def fit(self, max_steps: Optional[int] = None, log_every: int = 0):
        logged = []
        for epoch in range(self.config["max_epoch"]):
            for it in range(self.steps_per_epoch):
                if max_steps is not None and self.global_step >= max_steps:
                    return logged
                losses = self.step(epoch, it)
                if log_every and self.global_step % log_every == 0:
                    logged.append((self.global_step, [None if l is None else float(l) for l in losses]))
        return logged

    @torch.no_grad()
    def predict_all(self, chunk: int = 1 << 17) -> torch.Tensor:
        """[N,2] reconstruction: each point from the last ring that contains it (train_clustering.py:199-211);
        ranks contribute their rings' rows and one SUM all-reduce assembles the whole."""
        rec = torch.zeros(self.n, 2, device=self.device)
        win = winning_ring(self.dist, self.radii)
        for lo in range(0, self.n, chunk):
            hi = min(lo + chunk, self.n)
            x = self._inputs(lo, hi)
            for i in self.owned:
                sel = win[lo:hi] == i
                if bool(sel.any()):
                    o = self.engines[i].forward(x, self.enc_B, save=False)
                    rec[lo:hi][sel] = o[sel]
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.pg)
        return rec

    @torch.no_grad()
    def evaluate(self) -> float:
        in_image_space = bool(self.config.get("transform", False))
        ref = reconstruct(self.image, self.shape, in_image_space)
        return float(psnr(ref, reconstruct(self.predict_all(), self.shape, in_image_space)))

    def checkpoints(self) -> dict:
        """{ring: {'net', 'enc'}} of the owned rings (submodel_%d files of train_clustering.py:243-249)."""
        return {i: {"net": self.models[i].state_dict(), "enc": self.encoder.B} for i in self.owned}
