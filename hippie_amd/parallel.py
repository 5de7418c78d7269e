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


def backward_allreduce(engine, group=None, use_graph=True):
    """Backward + gradient mean over ranks: the whole active arena in one collective after the backward pass, on the
    communicator's stream.  What overlaps it is the OTHER model: the wave and the time cVAE step on two HIP streams, so one
    model's all-reduce runs under the other's kernels (north_star: "overlapped with the backward encoder GEMM").  Round 2
    also had a two-bucket form that reduced the decoder-side half of ONE model's gradients on a side stream under that model's
    encoder-side chain; the side stream's grid-filling weight-gradient launch stalls every small kernel of the chain behind
    its workgroups (linear_bwd_x 6 -> 32 us, profiles/r03_overlap_timeline.txt): 90.7 k vs 115.7 k samples/s.  Removed."""
    e = engine
    e.backward(use_graph)
    allreduce_mean_(e.grads[: e.plan.n_active], group)


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
