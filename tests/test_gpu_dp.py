"""GPU: the data-parallel step with two processes on the ONE available GPU (gloo backend moving CUDA
tensors — RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Checks the plumbing
bench.py relies on: graph replay -> gradient mean-all-reduce -> fused AdamW, replicas staying identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hippie_amd import parallel, planner
    from hippie_amd.engine import Engine
    from oracle import cvae_oracle as O
    z, L, B = 10, 50, 16
    eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=1e-4, clip=1.0, split_backward=True))
    om = O.OracleModel("unimodal", z, L, salt=rank)          # different init per rank: broadcast must fix it
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    dp = parallel.DataParallelEngine(eng, overlap=True)
    n = 64
    x, src, cls, eps = O.synth_inputs(n, L, z, salt=5)
    ok_grad = True
    for step in range(3):
        idx = parallel.shard_indices(n, rank, world, epoch=step, seed=1)[:B]
        eng.set_inputs(x[idx].cuda(), src[idx].cuda(), None, eps[idx].cuda())
        eng.forward(True, True)
        eng.backward(True)
        local = eng.grads.clone()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        parallel.allreduce_mean_(eng.grads[: eng.plan.n_active], None, 2)
        want = (gathered[0] + gathered[1]) / 2
        ok_grad &= bool(torch.allclose(eng.grads[: eng.plan.n_active], want[: eng.plan.n_active], rtol=1e-6, atol=1e-8))
        # the overlapped, two-bucket path (side stream + all-reduce under the encoder-side chain) gives the same
        # mean up to the summation order of the fp32 atomics in the weight-gradient GEMMs
        eng.forward(True, True)              # the backward pass consumes its inputs in place: stage them again
        parallel.backward_allreduce(eng, None, True, overlap=True)
        torch.cuda.synchronize()
        got, ref = eng.grads[: eng.plan.n_active], want[: eng.plan.n_active]
        ok_grad &= bool((got - ref).abs().max() <= 2e-5 * ref.abs().max())
        eng.optimizer_step(True)
    loss = dp.train_step(use_graph=True).clone()
    torch.cuda.synchronize()
    q.put((rank, ok_grad, eng.params.double().sum().item(), eng.params.double().abs().sum().item(), eng.adam_step, float(loss[0])))
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_on_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), "all-reduced gradients != mean of the ranks' local gradients"
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3], "replicas diverged"
    assert res[0][4] == res[1][4] == 4
    assert np.isfinite(res[0][5]) and np.isfinite(res[1][5])
