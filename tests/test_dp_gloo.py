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
