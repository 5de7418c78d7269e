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
    eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=1e-4, clip=1.0))
    om = O.OracleModel("unimodal", z, L, salt=rank)          # different init per rank: broadcast must fix it
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    dp = parallel.DataParallelEngine(eng)
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
        # backward_allreduce (what DataParallelEngine and bench.py call) gives the same mean up to the summation order of
        # the fp32 atomics in the weight-gradient GEMMs
        eng.forward(True, True)              # one backward per forward (the BN-backward statistic slots are zeroed by the forward): run it again
        parallel.backward_allreduce(eng, None, True)
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


# ---- DDP through the class surface: Trainer(strategy="ddp") -----------------------------------------------------------
class _ShardableLoader:
    """The protocol Trainer._shard looks for (scripts/pretrain_pipeline.py's loaders implement it): DistributedSampler
    semantics, per-rank batches of `batch`."""

    def __init__(self, x, labels, batch):
        self.x, self.labels, self.batch = x, labels, batch

    def __iter__(self):
        for i in range(0, len(self.x), self.batch):
            yield self.x[i: i + self.batch], self.labels[i: i + self.batch]

    def shard(self, rank, world, epoch, seed=0):
        from hippie_amd.parallel import shard_indices
        idx = shard_indices(len(self.x), rank, world, epoch=epoch, seed=seed)
        for i in range(0, len(idx), self.batch):
            j = idx[i: i + self.batch]
            yield self.x[j], self.labels[j]


def _eps_of(x, z):
    """reparameterisation noise as a fixed function of the batch itself: the oracle in the parent process reproduces it"""
    return torch.sin(997.0 * x.reshape(x.shape[0], -1)[:, :z].float().cpu()) * 1.3


def _ddp_trainer_worker(rank, world, port, q, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    from hippie_amd.trainer import Trainer
    from oracle import cvae_oracle as O
    z, L, B, n = 10, 50, 8, 32
    torch.manual_seed(100 + rank)                       # different initial weights per rank: the start-up broadcast must fix it
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    if rank == 0:
        sd0 = {k: v.numpy().copy() for k, v in net._pending_sd.items()}
    net.set_eps_source(lambda eng: _eps_of(eng.io("x"), z).to(eng.device))
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-5, weight_decay=0.01)
    x, src, cls, _ = O.synth_inputs(n, L, z, salt=21)
    loader = _ShardableLoader(x, src, B)
    val = _ShardableLoader(x[:16], src[:16], B)
    tr = Trainer(max_epochs=2, gradient_clip_val=1.0, strategy="ddp", default_root_dir=os.path.join(tmp, "ckpt"),
                 logger_path=os.path.join(tmp, "log.jsonl"), num_sanity_val_steps=0, seed=3)
    tr.fit(mod, loader, val)
    assert tr.world_size == world and net.dp_world == 1          # restored after fit
    eng = net._any_engine()
    exists = os.path.exists(tr.best_model_path)
    ck = torch.load(tr.best_model_path, weights_only=False) if exists else None
    q.put((rank, eng.params.double().sum().item(), eng.params.double().abs().sum().item(), eng.adam_step, tr.best_model_path, exists,
           # (numpy, not tensors: a tensor in a queue is handed over through the SENDER's shared-memory server, which is gone
           # once this process exits)
           {k: v.cpu().numpy() for k, v in net.state_dict().items()} if rank == 0 else None, sd0 if rank == 0 else None,
           sorted(ck.keys()) if ck else None, [h["val_loss"] for h in tr.history]))
    dist.destroy_process_group()


def test_trainer_ddp_two_ranks_equals_two_single_rank_oracles(tmp_path):
    """Trainer(strategy="ddp") under a 2-rank process group (gloo moving CUDA tensors; both ranks on the one GPU): start-up
    broadcast, DistributedSampler sharding, per-rank BatchNorm statistics, gradient MEAN between loss.backward() and
    optimizer.step(), rank-0 checkpoint that every rank can load — against two float64 oracles that each take their rank's
    shard, average their gradients and apply the same AdamW step (what Lightning-DDP does to the reference's module)."""
    import re
    from hippie_amd.parallel import shard_indices
    from oracle import cvae_oracle as O
    from tests import helpers as H
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_trainer_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
    r0, r1 = res
    assert r0[1] == r1[1] and r0[2] == r1[2], "replicas diverged"
    z, L, B, n, lr = 10, 50, 8, 32, 1e-5
    steps = 2 * (n // world // B)
    assert r0[3] == r1[3] == steps
    assert r0[4] == r1[4] and r0[5] and r1[5], "every rank must be able to load rank 0's checkpoint"
    assert r0[8] == ["epoch", "global_step", "optimizer_states", "state_dict"]
    assert r0[9] == r1[9], "ranks disagree on the monitored val_loss (checkpoint / early-stop decisions would diverge)"
    assert len([f for f in os.listdir(tmp_path / "ckpt")]) >= 1 and os.path.exists(tmp_path / "log.jsonl")
    # ---- two float64 oracles, one per rank, gradients averaged
    x, src, cls, _ = O.synth_inputs(n, L, z, salt=21)
    oms = [O.OracleModel("unimodal", z, L, dtype=torch.float64) for _ in range(world)]
    for om in oms:
        om.load({k: torch.from_numpy(v) for k, v in r0[7].items()})           # rank 0's constructor initialisation, broadcast to all
        for k in om.state:                                                     # BatchNorm buffers of a fresh module
            if k.endswith("running_mean") or k.endswith("num_batches_tracked"):
                om.state[k].zero_()
            elif k.endswith("running_var"):
                om.state[k].fill_(1.0)
    for epoch in range(2):
        shards = [shard_indices(n, r, world, epoch=epoch, seed=3) for r in range(world)]
        for i in range(0, n // world, B):
            grads = []
            for r, om in enumerate(oms):
                j = shards[r][i: i + B]
                xb = x[j]
                for k in om.param_keys:
                    om.state[k].grad = None
                outs = om.forward((xb.double(), src[j], None), _eps_of(xb, z).double(), True)
                om.losses((xb.double(), src[j], None), outs, 1.0)[0].backward()
                grads.append(om.grads())
            with torch.no_grad():
                mean = {k: (None if grads[0][k] is None else (grads[0][k] + grads[1][k]) / 2) for k in grads[0]}
                for om in oms:
                    g = {k: (None if v is None else v.clone()) for k, v in mean.items()}
                    O.clip_grad_norm(list(g.values()), 1.0)
                    for k in om.param_keys:
                        if g[k] is not None and k not in om.exp_avg:
                            om.exp_avg[k] = torch.zeros_like(om.state[k])
                            om.exp_avg_sq[k] = torch.zeros_like(om.state[k])
                    om.step_count += 1
                    O.adamw_step({k: om.state[k] for k in om.param_keys}, g, om.exp_avg, om.exp_avg_sq, om.step_count, lr, 0.01)
    sd = r0[6]
    for k in oms[0].param_keys:
        if mean[k] is None:
            continue
        a, b = sd[k].astype(np.float64).reshape(-1), oms[0].state[k].detach().numpy().reshape(-1)
        assert np.abs(a - b).max() <= 2.2 * lr * steps + 1e-7, k
        if re.search(H.ZERO_GRAD_RE, k):
            continue
        # UNMASKED comparison at 8 rows per rank: besides Adam's +-lr on noise-level gradients, a leaky-ReLU sign that differs
        # from the float64 oracle's changes upstream gradients by a few per cent, i.e. the Adam update by a few per cent of lr
        # per step.  A wrong shard, synced statistics or a missing all-reduce would move MOST elements by ~lr per step.
        gk = np.abs(mean[k].numpy().reshape(-1))
        bad = (np.abs(a - b) > 1e-4 * np.abs(b) + 0.25 * lr) & (gk > 1e-2 * gk.max())
        assert bad.mean() <= 2e-2, f"{k}: {bad.sum()} of {bad.size} elements with significant gradient differ by more than a quarter step"
    for k, v in sd.items():          # rank 0's BatchNorm running statistics are the ones checkpointed: they follow rank 0's shards
        if "running_" in k:
            # (statistics of 8 units per rank: fp32 rounding in the convs moves them by 0.5-1.1e-4 of the tensor's max on either fp32 matrix
            # path — 1.07e-4 measured for decoder.layer4.1.bn1 on the three-term path, 0.9e-4 on the fp32 cores)
            H.assert_close(v, oms[0].state[k].detach().numpy(), 2e-4, k)


# ---- the RCCL ("nccl") backend inside the suite: world size 1 -------------------------------------------------------------
def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPPIE_FORCE_DIST="1", GPU_MAX_HW_QUEUES="8")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from hippie_amd import parallel, planner
    from hippie_amd.engine import Engine
    from oracle import cvae_oracle as O
    z, B, steps = 10, 32, 3
    out = {}
    for name, use_dp, bucketed in (("plain", False, False), ("rccl", True, False), ("rccl_bucketed", True, True)):
        engs, streams = [], []
        for k, (L, clip) in enumerate(((50, 0.0), (100, 1.0))):
            e = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=1e-4, clip=clip, deterministic_wgrad=True, bucketed_bwd=bucketed))
            assert (e.plan.grad_buckets is not None) == bucketed
            om = O.OracleModel("unimodal", z, L, salt=30 + k)
            e.load_state_dict({kk: v.detach() for kk, v in om.state.items()})
            x, src, cls, eps = O.synth_inputs(B, L, z, salt=30 + k)
            e.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
            engs.append(parallel.DataParallelEngine(e, dist.group.WORLD) if use_dp else e)
            streams.append(torch.cuda.Stream(device=dev))
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(cur)
        # the bench's structure: the wave and the time model on two HIP streams, graph replay, ONE communicator
        for _ in range(steps):
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.train_step(use_graph=True)
        for s in streams:
            cur.wait_stream(s)
        torch.cuda.synchronize()
        raw = [e.engine if use_dp else e for e in engs]
        out[name] = [(e.params.cpu().numpy().copy(), e.grads.cpu().numpy().copy(), e.scalars(), e.adam_step) for e in raw]
    q.put((out, dist.get_backend(), dist.get_world_size()))
    dist.destroy_process_group()


def test_rccl_world1_data_parallel_step_equals_plain_step_bit_for_bit():
    """The first RCCL run inside the suite: DataParallelEngine.train_step (hipGraph replay of fwd / bwd / opt with
    ncclAllReduce(AVG) of the gradient arena between bwd and opt) for the wave and the time model on two HIP streams over
    ONE communicator, world size 1 — against the same engines stepping without torch.distributed.  The mean over one rank is
    the identity and deterministic_wgrad orders every sum, so parameters, gradients and losses must agree BIT FOR BIT.  Also in the
    bucketed form (backward in two halves, the decoder-side bucket reduced asynchronously under the encoder-side half)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out, backend, world = q.get(timeout=600)
    p.join(60)
    assert backend == "nccl" and world == 1
    for variant in ("rccl", "rccl_bucketed"):      # one collective after the pass | two halves, two asynchronous buckets (TrainCfg.bucketed_bwd)
        for k in range(2):
            pa, ga, sa, ta = out["plain"][k]
            pb, gb, sb, tb = out[variant][k]
            assert ta == tb == 3
            np.testing.assert_array_equal(ga, gb, err_msg=f"{variant} model {k}: gradients differ")
            np.testing.assert_array_equal(pa, pb, err_msg=f"{variant} model {k}: parameters differ")
            assert sa == sb, (variant, sa, sb)


def _pipeline_rank(rank, world, port, data_root, out_dir, q, model_type="unimodal"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HIPPIE_SINGLE_DEVICE="1", HIPPIE_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import pretrain_pipeline as pp
    paths = pp.main(["--dataset", "cellexplorer-celltype", "--data-root", data_root, "--output-dir", out_dir, "--batch-size", "32",
                     "--pretrain-max-epochs", "2", "--finetune-max-epochs", "1", "--z_dim", "5", "--learning-rate", "1e-4", "--strategy", "ddp",
                     "--model-type", model_type])
    q.put((rank, paths))
    dist.destroy_process_group()


@pytest.mark.parametrize("model_type", ["unimodal", "multimodal"])
def test_pipeline_script_runs_data_parallel_under_a_launcher(tmp_path, model_type):
    """BASELINE configs[3]'s route (multimodal: configs[4]'s): `torchrun --nproc-per-node N scripts/pretrain_pipeline.py` — here two ranks started by hand (gloo,
    both on the one GPU): the script initialises the process group itself, shards its loaders, trains both models data-parallel
    (pretrain -> rank-0 checkpoint reload on every rank -> fine-tune), and rank 0 alone writes logs, checkpoints and the CSVs."""
    import pandas as pd
    from tests.test_gpu_pipeline import make_root
    data = tmp_path / "datasets"
    data.mkdir()
    make_root(data, np.random.default_rng(5))
    out = tmp_path / "out"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_rank, args=(r, world, port, str(data), str(out), q, model_type)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0] == res[1]
    multimodal = model_type == "multimodal"
    for name in (("joint",) if multimodal else ("waveform", "isi", "joint")):
        df = pd.read_csv(res[0][name])
        emb = np.stack([np.array(v.strip("[]").split(), dtype=float) for v in df["embeddings"]])
        assert np.isfinite(emb).all() and emb.shape[1] == (5 if multimodal or name != "joint" else 10)
    logs = sorted(f for f in os.listdir(out) if f.endswith("_log.jsonl"))
    assert logs == (["joint_finetune_log.jsonl", "joint_pretrain_log.jsonl"] if multimodal else
                    ["time_finetune_log.jsonl", "time_pretrain_log.jsonl", "wave_finetune_log.jsonl", "wave_pretrain_log.jsonl"])
    import json
    rec = [json.loads(ln) for ln in open(out / ("joint_pretrain_log.jsonl" if multimodal else "wave_pretrain_log.jsonl"))]
    assert len(rec) == 2 and all(r["world_size"] == 2 for r in rec)            # written once (rank 0), two epochs
