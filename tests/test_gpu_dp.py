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
        eng.forward(True, True)              # one backward per forward (the BN-backward statistic slots are zeroed by the forward): run it again
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


def _sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hippie_amd import parallel, planner
    from hippie_amd.engine import Engine
    from oracle import cvae_oracle as O
    z, L, B = 10, 50, 8
    eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=1e-4, sync_bn_world=world))
    om = O.OracleModel("unimodal", z, L, salt=3)
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    x, src, cls, eps = O.synth_inputs(world * B, L, z, salt=9)
    sl = slice(rank * B, (rank + 1) * B)
    eng.set_inputs(x[sl].cuda(), src[sl].cuda(), None, eps[sl].cuda())
    enc, mu, lv, rec = eng.forward(True)
    eng.backward()
    parallel.allreduce_mean_(eng.grads[: eng.plan.n_active])
    torch.cuda.synchronize()
    from tests import helpers as H
    q.put((rank, enc.cpu().numpy(), rec.cpu().numpy(), {k: v.cpu().numpy() for k, v in eng.grad_dict().items()},
           {k: v.cpu().numpy() for k, v in eng.state_dict().items() if "running" in k},
           {k: m.numpy() for k, m in H.engine_masks(eng).items()}))
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_global_batch_oracle():
    """Two ranks x 8 rows with sync-BatchNorm against the float64 oracle at 16 rows: forward rows, running
    statistics and the all-reduced gradients (the DP path's parity test proper; criterion of tests/helpers.parity)."""
    import re
    from oracle import cvae_oracle as O
    from tests import helpers as H
    world, z, L, B = 2, 10, 50, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
    x, src, cls, eps = O.synth_inputs(world * B, L, z, salt=9)
    oms = [O.OracleModel("unimodal", z, L, salt=3, dtype=dt) for dt in (torch.float32, torch.float64)]
    outs = []
    # the oracle runs the GLOBAL batch on the leaky-ReLU branches the two ranks took on their halves (tests/helpers.py)
    masks = {k: torch.from_numpy(np.concatenate([r[5][k] for r in res], axis=0)) for k in res[0][5]}
    for om, dt in zip(oms, (torch.float32, torch.float64)):
        o = om.forward((x.to(dt), src, None), eps.to(dt), True, masks=masks)
        om.losses((x.to(dt), src, None), o)[0].backward()
        outs.append(o)
    n = lambda t: t.detach().numpy()
    for rank, enc, rec, grads, running, _ in res:
        sl = slice(rank * B, (rank + 1) * B)
        H.parity(enc, n(outs[0][0])[sl], n(outs[1][0])[sl], f"rank {rank} enc")
        H.parity(rec, n(outs[0][3])[sl], n(outs[1][3])[sl], f"rank {rank} rec")
        for k, v in running.items():
            np.testing.assert_allclose(v, oms[0].state[k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
        g32, g64 = oms[0].grads(), oms[1].grads()
        for k, g in g32.items():
            if g is None or re.search(H.ZERO_GRAD_RE, k):
                continue
            H.parity(grads[k], g.numpy(), g64[k].numpy(), f"rank {rank} grad {k}")
    for k in res[0][3]:
        np.testing.assert_array_equal(res[0][3][k], res[1][3][k])        # identical after the all-reduce


def test_bench_self_launches_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher: the parent must start the ranks itself (before touching the
    GPU), forward exactly one JSON line and report the communicator's world size.  Rehearsed on the one GPU of
    the test box with gloo (RCCL refuses two ranks per device)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HIPPIE_SINGLE_DEVICE="1", HIPPIE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["config"]["global_batch"] == 1024
    assert out["value"] > 0 and np.isfinite(out["config"]["final_loss_wave"])
    # a failing rank makes the parent fail
    env["HIPPIE_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip()
