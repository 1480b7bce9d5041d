"""N>1 path on CPU: world_size-2 gloo run of the product's sharding + exchange step
(inr_mi355x.train.shard_rows / allreduce_step_outputs).  The kernel's per-shard output
(partial gradient sums divided by the GLOBAL count) is produced by the oracle here, because the
HIP engine needs a GPU; what is under test is that shard -> partial -> SUM all-reduce equals the
single-rank full-batch step, including ragged shards and masked rows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O

NET = dict(network_input_size=16, network_output_size=2, network_depth=3, network_width=32, last_tanh=True)
ENC = dict(embedding="gauss", scale=2, embedding_size=8, coordinates_size=3)


def _partial(sd, B_enc, coords, gt, mask, lo, hi, count):
    """What inr_train_step leaves on one rank: sum over its rows of d(loss_row)/d(params) with the
    loss normalised by the global count (inv_count = 1/count)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.siren_forward(params, O.encode(coords[lo:hi], B_enc, "gauss"), NET)
    g = gt[lo:hi]
    if mask is not None:
        m = mask[lo:hi]
        out, g = out[m], g[m]
    loss = 0.5 * torch.sum((out - g) ** 2) / (count * 2)
    grads = torch.autograd.grad(loss, list(params.values()))
    return torch.cat([x.reshape(-1) for x in grads]), loss.detach()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inr_mi355x.train import shard_rows, allreduce_step_outputs
    torch.manual_seed(0)
    B_enc = O.encoder_init(ENC)
    sd = O.init_siren(NET)
    g = torch.Generator().manual_seed(1)
    n = 1003
    coords = torch.rand(n, 3, generator=g) * 2 - 1
    gt = torch.randn(n, 2, generator=g) * 0.3
    mask = torch.rand(n, generator=g) < 0.5
    res = []
    for (lo, hi, use_mask) in ((0, 400, False), (400, 1003, False), (0, 1003, True), (1000, 1003, False)):
        m = mask if use_mask else None
        count = int(mask[lo:hi].sum()) if use_mask else hi - lo
        slo, shi = shard_rows(lo, hi, rank, world)
        grads, loss = _partial(sd, B_enc, coords, gt, m, slo, shi, count)
        if use_mask:  # the engine's layout: loss word behind the gradient, one collective
            gbuf = torch.cat([grads, torch.zeros(1)])
            grads = gbuf[:-1]
            loss = allreduce_step_outputs(grads, loss.reshape(1), world, gbuf=gbuf)
        else:
            loss = allreduce_step_outputs(grads, loss.reshape(1), world)
        full_g, full_l = _partial(sd, B_enc, coords, gt, m, lo, hi, count)
        res.append((float((grads - full_g).norm() / full_g.norm()), float(abs(loss - full_l) / abs(full_l))))
    q.put((rank, res))
    dist.destroy_process_group()


def test_dp_two_ranks_equals_single():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in out:
        for g_err, l_err in res:
            assert g_err < 2e-6 and l_err < 2e-6, (rank, res)


# ---- sharded update: reduce-scatter -> Adam on 1/N of the entries -> all-gather (inr_mi355x.engine.sharded_exchange) ----
def _flat_adam(p, g, st, lr):
    """the oracle's Adam on a flat slice (elementwise: a slice of the vector steps like the vector)"""
    O.adam_step({"p": p}, {"p": g}, st, lr)


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inr_mi355x.engine import sharded_exchange
    from inr_mi355x.train import shard_rows
    torch.manual_seed(0)
    B_enc = O.encoder_init(ENC)
    sd = O.init_siren(NET)
    shapes = [(k, v.shape, v.numel()) for k, v in sd.items()]
    flat = torch.cat([v.reshape(-1) for v in sd.values()]).clone()
    P = flat.numel()
    chunk = -(-(P + 1) // world)
    ref_flat = flat.clone()
    g = torch.Generator().manual_seed(1)
    n = 1003
    coords = torch.rand(n, 3, generator=g) * 2 - 1
    gt = torch.randn(n, 2, generator=g) * 0.3
    lo_own, hi_own = min(rank * chunk, P), min(rank * chunk + chunk, P)
    st = {"p": {"step": 0, "m": torch.zeros(hi_own - lo_own), "v": torch.zeros(hi_own - lo_own)}}
    st_ref = {"p": {"step": 0, "m": torch.zeros(P), "v": torch.zeros(P)}}
    gbuf, gchunk, pchunk, pgather = torch.zeros(chunk * world), torch.zeros(chunk), torch.zeros(chunk), torch.zeros(chunk * world)
    errs = []

    def unflat(v):
        out, o = {}, 0
        for k, shp, cnt in shapes:
            out[k] = v[o:o + cnt].reshape(shp)
            o += cnt
        return out

    for (lo, hi) in ((0, 400), (400, 1003), (1000, 1003)):  # the last batch leaves a rank of three without rows
        slo, shi = shard_rows(lo, hi, rank, world)
        gbuf.zero_()
        if shi > slo:
            grads, loss = _partial(unflat(flat), B_enc, coords, gt, None, slo, shi, hi - lo)
            gbuf[:P], gbuf[P] = grads, loss

        def update(a, b, gc):
            assert (a, b) == (lo_own, hi_own)
            if b > a:
                _flat_adam(flat[a:b], gc[:b - a], st, 1e-3)

        loss = sharded_exchange(gbuf, gchunk, pchunk, pgather, flat, rank, world, None, update)
        full_g, full_l = _partial(unflat(ref_flat), B_enc, coords, gt, None, lo, hi, hi - lo)
        _flat_adam(ref_flat, full_g, st_ref, 1e-3)
        errs.append((float((flat - ref_flat).abs().max()), float(abs(loss - full_l) / abs(full_l))))
    q.put((rank, errs, flat.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_update_equals_single(world):
    """P + 1 = 1667 words: chunks of 834 on two ranks, 556 on three (one word of padding each); after every step the
    replicas hold bitwise-equal parameters, equal to the single-rank Adam trajectory up to the summation order of the
    gradient (Adam's first steps move an entry by ~lr whatever its gradient: the bound is a fraction of that)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, errs, flat in out:
        for p_err, l_err in errs:
            assert p_err < 2e-5 and l_err < 2e-6, (rank, errs)
        np.testing.assert_array_equal(flat, out[0][2])
