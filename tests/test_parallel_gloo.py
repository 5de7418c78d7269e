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


class _InterpEngine:
    """What parallel.backward_allreduce needs of an Engine, executed by the numpy interpreter on CPU arenas: the host logic of the
    bucketed backward (segment halves, bucket ranges, asynchronous collectives) without a GPU."""

    def __init__(self, plan):
        from tests import helpers as H
        from oracle import interp
        self.plan, self.train_cfg, self._interp = plan, plan.train, interp
        self.ops = plan.ops.array()
        self.A = H.make_arenas(plan)
        self.grads = torch.from_numpy(self.A.mem[2].view(np.float32))        # HP_SPACE_GRAD, zero-copy

    def run(self, seg, use_graph=False):
        self._interp.run(self.ops, self.A, *self.plan.ops.segments[seg])

    def backward(self, use_graph=False):
        self.run("bwd")


def _bucket_worker(rank, world, port, out, kind):
    from hippie_amd import planner
    from tests import helpers as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    z, L, L2, B = 10, 50, 100, 4
    multi = kind == "multimodal"
    cfg = planner.ModelCfg(kind=kind, z_dim=z, output_size=L, output_size2=L2) if multi else planner.ModelCfg(kind=kind, z_dim=z, output_size=L)
    x, src, cls, eps = O.synth_inputs(world * B, L, z, salt=3)
    x2 = O.synth_inputs(world * B, L2, z, salt=4)[0]
    om = O.OracleModel(kind, z, L, output_size2=L2, salt=3) if multi else O.OracleModel(kind, z, L, salt=3)
    sl = slice(rank * B, (rank + 1) * B)
    got = []
    for bucketed in (True, False):
        plan = planner.lower(cfg, B, planner.TrainCfg(lr=1e-3, clip=1.0, bucketed_bwd=bucketed, deterministic_wgrad=True), with_class=True)
        e = _InterpEngine(plan)
        H.load_state(plan, e.A, om.state)
        H.set_io(plan, e.A, "x", x[sl].numpy())
        if multi:
            H.set_io(plan, e.A, "x2", x2[sl].numpy())
        H.set_io(plan, e.A, "src", src[sl].numpy()), H.set_io(plan, e.A, "cls", cls[sl].numpy()), H.set_io(plan, e.A, "eps", eps[sl].numpy())
        e.run("fwd_train")
        if bucketed:
            s_ = plan.ops.segments
            assert s_["bwd_dec"][0] == s_["bwd"][0] and sum(s_["bwd_dec"]) == s_["bwd_enc"][0] and sum(s_["bwd_enc"]) == sum(s_["bwd"])
            cover = sorted(r for half in plan.grad_buckets for r in half)
            assert cover[0][0] == 0 and cover[-1][1] == plan.n_active and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        else:
            assert plan.grad_buckets is None and "bwd_dec" not in plan.ops.segments
        local = None
        if bucketed:
            e2 = _InterpEngine(plan)                       # the same rank's gradients without any collective, for the mean check below
            e2.A.mem[0][:], e2.A.mem[1][:], e2.A.mem[3][:] = e.A.mem[0], e.A.mem[1], e.A.mem[3]
            e2.backward()
            local = e2.grads[: plan.n_active].clone()
        parallel.backward_allreduce(e, None, False)
        e.run("opt")
        got.append((e.grads[: plan.n_active].clone(), torch.from_numpy(e.A.mem[1].view(np.float32).copy())))
        if bucketed:
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            assert torch.equal(got[0][0], sum(gathered) / world), "bucketed all-reduce != mean of the ranks' gradients"
    same = torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    out.put((rank, bool(same), float(got[0][0].abs().sum())))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["unimodal", "multimodal"])
def test_bucketed_backward_allreduce_equals_unbucketed_world2(kind):
    """TrainCfg(bucketed_bwd=True): "bwd" split into "bwd_dec" | "bwd_enc" with the decoder-side gradient range all-reduced while the
    second half runs.  Two gloo ranks, each on its own shard, programs executed by the numpy interpreter: gradients after the collective and
    parameters after AdamW (+ clip) are BIT-EQUAL to the unbucketed plan's (one collective after the whole pass), the halves tile the
    pass, the buckets tile the active arena, and the result is the mean of the ranks' local gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res) and all(r[2] > 0 for r in res)
    assert res[0][2] == res[1][2]                      # both ranks hold the same reduced gradients


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
