"""Data-parallel pieces of the pretrain hot path: one process per GPU, replicated parameters, per-rank
batches (Lightning-DDP semantics: `DataLoader(batch_size=512)` is per process, BatchNorm statistics stay
local), gradients mean-all-reduced over RCCL (xGMI) between the backward and the fused AdamW launch.
The reference has no distributed code of its own; these are the semantics Lightning's default strategy
would give its scripts (scripts/train_model_with_multimodal.py:200-207)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_indices(n, rank, world, epoch=0, seed=0, shuffle=True, drop_last=False):
    """DistributedSampler semantics: a seeded per-epoch permutation, padded (by wrapping) to a multiple of
    `world`, rank r takes positions r::world.  Every rank gets the same count; together they cover all n."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        perm = torch.randperm(n, generator=g)
    else:
        perm = torch.arange(n)
    if drop_last:
        total = n - n % world
        perm = perm[:total]
    else:
        total = -(-n // world) * world
        if total > n:
            perm = torch.cat([perm, perm[: total - n]])
    return perm[rank:total:world]


_NCCL_AVG = [True]


def allreduce_mean_(flat_grads, group=None, buckets=1):
    """In-place mean over ranks of a flat gradient arena (works on any backend: nccl = RCCL on ROCm,
    gloo on CPU).  `buckets` > 1 splits the arena so that the collective of one bucket can overlap
    whatever is enqueued next on other streams."""
    world = dist.get_world_size(group)
    if world == 1 and not os.environ.get("HIPPIE_FORCE_DIST"):     # (the env knob keeps the 1-rank collective for measurements)
        return flat_grads
    n = flat_grads.numel()
    step = -(-n // buckets)
    backend = dist.get_backend(group)
    for lo in range(0, n, step):
        chunk = flat_grads[lo: lo + step]
        if backend == "nccl" and _NCCL_AVG[0]:
            try:
                dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=group)
                continue
            except RuntimeError:                      # an RCCL build without ncclAvg: sum and scale instead
                _NCCL_AVG[0] = False
        dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=group)
        chunk.div_(world)
    return flat_grads


def broadcast_(tensors, src=0, group=None):
    """Make every rank start from rank `src`'s parameters / buffers / optimiser state."""
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def backward_allreduce(engine, group=None, use_graph=True, comm_stream=None):
    """Backward + gradient mean over ranks.

    Plans lowered with TrainCfg(bucketed_bwd=True) (what Trainer's DDP strategy, bench.py --gpus N and the pipeline under torchrun
    use): the pass runs as its two halves.  After "bwd_dec" the decoder-side gradient range is complete and its all-reduce is issued
    on `comm_stream` while "bwd_enc" (heads, encoder input-gradients, the encoder's weight-gradient group) runs on the model's OWN
    stream; the second bucket follows it, and the model's stream waits for both before the optimiser.  SURVEY section 8(e): "bucket A
    = decoder + heads, reduced while the encoder backward GEMMs run".  All compute stays on one stream: round 2's form moved the
    decoder-side weight-gradient group to a side stream, whose grid-filling launch stalled every small kernel of the main chain
    behind its workgroups (profiles/r03_overlap_timeline.txt, 90.7 k vs 115.7 k samples/s).

    comm_stream: the stream the collectives run on (torch issues a synchronous collective on the CURRENT stream).  None: a side stream
    probed not to share a hardware queue with the model's stream (streams.pick_side_stream, cached) — on a shared queue the side
    stream's event waits are barriers in front of the model's own kernels (8.3 instead of 4.3 ms per pair-step at one rank).  Callers
    that step several models side by side pass one comm stream (and one process group) per model: bench.py.

    Other plans (and sync-BatchNorm runs, which are collective-bound anyway): the whole active arena in one collective after the pass,
    on the model's stream."""
    e = engine
    buckets = getattr(e.plan, "grad_buckets", None)
    forced = bool(os.environ.get("HIPPIE_FORCE_DIST"))     # (the env knob keeps the 1-rank collective for measurements)
    if buckets is None or (dist.get_world_size(group) == 1 and not forced) or e.train_cfg.sync_bn_world > 1:
        e.backward(use_graph)
        allreduce_mean_(e.grads[: e.plan.n_active], group)
        return
    if not e.grads.is_cuda:                # (CPU arenas: the numpy interpreter under gloo, tests/test_parallel_gloo.py — same halves, same buckets)
        for seg, ranges in zip(("bwd_dec", "bwd_enc"), buckets):
            e.run(seg, use_graph)
            for lo, hi in ranges:
                allreduce_mean_(e.grads[lo:hi], group)
        return
    cur = torch.cuda.current_stream(e.grads.device)
    from .program import debug_knob
    variant = debug_knob("HIPPIE_DP_VARIANT", "")          # (measurement only: which part of the two-bucket form costs what)
    if variant == "whole":                                  # the split program, one collective after both halves
        e.run("bwd_dec", use_graph), e.run("bwd_enc", use_graph)
        allreduce_mean_(e.grads[: e.plan.n_active], group)
        return
    if comm_stream is None:
        from .streams import pick_side_stream
        comm_stream = pick_side_stream([cur], e.grads.device)
    if variant == "samestream":
        comm_stream = cur
    for seg, ranges in zip(("bwd_dec", "bwd_enc"), buckets):
        e.run(seg, use_graph)
        if comm_stream is not cur:
            comm_stream.wait_stream(cur)
        with torch.cuda.stream(comm_stream):
            for lo, hi in ranges:
                if variant != "nocoll":
                    allreduce_mean_(e.grads[lo:hi], group)
    if comm_stream is not cur:
        cur.wait_stream(comm_stream)


class DataParallelEngine:
    """Wraps an Engine: train_step = forward + backward + all-reduce(mean) + AdamW."""

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group
        broadcast_([engine.params, engine.bufs, engine.m, engine.v], 0, group)

    def train_step(self, use_graph=True):
        e = self.engine
        e.forward(True, use_graph)
        backward_allreduce(e, self.group, use_graph)
        e.optimizer_step(use_graph)
        return e.io("scalars")
