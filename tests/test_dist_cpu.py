"""world_size-2 gloo tests (CPU) of the data-parallel host logic: bucketed averaging of a flat
gradient buffer, bf16 wire compression, branch agreement, batch sharding.  The HIP kernels are
not involved (they need a GPU); this covers the N>1 path's bookkeeping by construction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xggm_amd.dist import GradSync, sync_branch, shard_batch, ranges_to_buckets
        n = 10_000
        base = torch.arange(n, dtype=torch.float32)
        g = base * (rank + 1)              # rank r holds (r+1) * base
        ranges = [(8, 1000), (2000, 9992)]  # two "active groups"; the rest must stay untouched
        gs = GradSync(g, bucket_elems=1024)
        gs.sync(ranges)
        mean = base * (sum(range(1, world + 1)) / world)
        ok = True
        for s, e in ranges:
            ok &= torch.allclose(g[s:e], mean[s:e])
        ok &= torch.equal(g[:8], base[:8] * (rank + 1)) and torch.equal(g[1000:2000], base[1000:2000] * (rank + 1))
        # bf16 on the wire: result within bf16 rounding of the mean
        g2 = base * (rank + 1)
        GradSync(g2, wire_dtype=torch.bfloat16, bucket_elems=4096).sync([(0, n)])
        ok &= bool(((g2 - mean).abs() <= 8e-3 * mean.abs() + 1e-6).all())
        # many small buckets: both wire buffers are reused several times; untouched gaps stay untouched
        g3 = base * (rank + 1)
        GradSync(g3, wire_dtype=torch.bfloat16, bucket_elems=700).sync(ranges)
        for s, e in ranges:
            ok &= bool(((g3[s:e] - mean[s:e]).abs() <= 8e-3 * mean[s:e].abs() + 1e-6).all())
        ok &= torch.equal(g3[:8], base[:8] * (rank + 1)) and torch.equal(g3[1000:2000], base[1000:2000] * (rank + 1))
        # every rank follows rank 0's branch
        br = sync_branch(rank == 0, "cpu")
        ok &= (br is True)
        # sharding
        batch = {"x": torch.arange(8).view(8, 1), "sent": "keep"}
        sh = shard_batch(batch, rank, world)
        ok &= sh["x"].flatten().tolist() == list(range(rank * 4, rank * 4 + 4)) and sh["sent"] == "keep"
        ok &= ranges_to_buckets([(0, 10), (20, 25)], 4) == [(0, 4), (4, 8), (8, 10), (20, 24), (24, 25)]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_sync_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_noop():
    from xggm_amd.dist import GradSync
    g = torch.ones(16)
    GradSync(g).sync([(0, 16)])
    assert torch.equal(g, torch.ones(16))
