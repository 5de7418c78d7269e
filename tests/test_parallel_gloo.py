"""Multi-process (gloo, world_size 2) tests of the data-parallel host logic: index sharding and the
gradient mean-all-reduce, checked against the oracle's per-shard gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hippie_amd import parallel
from oracle import cvae_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    z, L, n = 10, 50, 12
    idx = parallel.shard_indices(n, rank, world, epoch=3, seed=7)
    x, src, cls, eps = O.synth_inputs(n, L, z, salt=1)
    m = O.OracleModel("unimodal", z, L, salt=1)
    outs = m.forward((x[idx], src[idx], None), eps[idx], True)
    m.losses((x[idx], src[idx], None), outs)[0].backward()
    keys = [k for k in m.param_keys if m.state[k].grad is not None]
    flat = torch.cat([m.state[k].grad.reshape(-1) for k in keys])
    local = flat.clone()
    parallel.allreduce_mean_(flat, buckets=3)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok = torch.allclose(flat, want, rtol=1e-6, atol=1e-7)
    p = torch.full((5,), float(rank))
    parallel.broadcast_([p], 0)
    out.put((rank, bool(ok), idx.tolist(), p.tolist()))
    dist.destroy_process_group()


def test_allreduce_mean_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    res.sort()
    assert all(r[1] for r in res)
    a, b = res[0][2], res[1][2]
    assert len(a) == len(b) == 6 and sorted(a + b) == list(range(12))
    assert res[1][3] == [0.0] * 5


def test_shard_indices_padding_and_determinism():
    for n, world in ((13, 4), (3797, 8), (512, 2)):
        parts = [parallel.shard_indices(n, r, world, epoch=1, seed=42) for r in range(world)]
        assert len({len(p) for p in parts}) == 1
        allidx = torch.cat(parts)
        assert set(allidx.tolist()) == set(range(n))
        assert len(allidx) == -(-n // world) * world
        again = parallel.shard_indices(n, 0, world, epoch=1, seed=42)
        assert torch.equal(parts[0], again)
        other = parallel.shard_indices(n, 0, world, epoch=2, seed=42)
        assert not torch.equal(parts[0], other)
    p = parallel.shard_indices(13, 1, 4, shuffle=False, drop_last=True)
    assert p.tolist() == [1, 5, 9]
