#!/usr/bin/env python3
"""Headline benchmark: pretrain samples/sec of the waveform + spike-timing cVAE pair at batch 512.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one optimisation step (forward, loss, backward, [clip], AdamW) of the wave model
(L=50, no clipping) AND of the time model (L=100, clip 1.0) on the same batch of 512 units — the
reference's default `--model-type unimodal` pipeline trains exactly this pair
(scripts/train_model_with_multimodal.py:200-224).  Inputs are synthetic tensors of the
cellexplorer-celltype pretrain shape (BASELINE.json configs[1]), resident in HBM before the timed
region; each step gathers its batch by index from them.  One process per GPU; with N > 1 every rank
holds a replica, takes its own 512-unit batch (Lightning-DDP semantics, weak scaling) and the
gradients are mean-all-reduced over RCCL between backward and AdamW.

Prints ONE JSON line (rank 0).  `dtype` "f32": the arithmetic is the reference's fp32; `matrix_path` says which matrix cores carry it —
"bf16x3" (default: every fp32 operand split exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulation;
error against fp64 at or below the fp32 matrix cores', tests/test_gpu_split.py) or "f32" (`--matrix-path f32`: v_mfma_f32_32x32x2_f32).
`roofline` is for the dominant kernel (the implicit-GEMM convolution `conv_taps_kernel`), timed with HIP events per launch in an untimed pass
right after the timed region — algorithmic fp32 flops against the path's peak (bf16 dense / 6 = 417 TFLOP/s; `frac_of_f32_mfma_peak`: against 157.3); `cpu_baseline` times the torch-CPU oracle on a bounded sample on this host's cores.
Secondary figures on the same line (N = 1): `trainer_samples_per_s` (the reference-API path: Trainer.fit of both modules,
concurrently), `inference_path` (get_embeddings over the pool), `dp_overhead_1rank` (the step with a 1-rank RCCL all-reduce in it);
`config.stream_pair` / `config.run_ahead` say how the two model streams were scheduled.
"""
import argparse
import json
import os
import sys
import time

# The step keeps three HIP streams busy (wave model, time model, RCCL).  ROCm multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4, shared with the null and copy streams); two streams on one queue
# serialise.  Measured with a 1-rank RCCL all-reduce in the step: 61.0k samples/s at 4 queues, 84.0k at 8.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hippie_amd import parallel, planner, program as P          # noqa: E402
from hippie_amd.engine import Engine                   # noqa: E402

PMC_SUMMARY = os.path.join(ROOT, "profiles", "r04b_conv_pmc.json")   # written by tools/pmc_summary.py from rocprofv3 --pmc passes
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_TFLOPS = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": 2500.0, "bf16x3": 2500.0 / 6}      # bf16x3: six bf16 products per fp32 product
#   (bf16: ~2.5 PFLOP/s dense, v_mfma_f32_32x32x16_bf16)
N_UNITS = 15631                   # cellexplorer-celltype pretrain pool, 80 % split
BATCH = 512
Z_DIM = 10


def synth_dataset(n, device, seed=42, lw=50, lt=100):
    """Synthetic waveform [n,50] / ISI [n,100] / source labels, SURVEY.md section 8(d)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.linspace(0, 1, lw)[None, :]
    ratio = 0.2 + 0.4 * torch.rand(n, 1, generator=g)
    wave = -torch.exp(-0.5 * ((t - 0.3) / 0.04) ** 2) + ratio * torch.exp(-0.5 * ((t - 0.5) / 0.1) ** 2)
    wave = wave + 0.02 * torch.randn(n, lw, generator=g)
    lo, hi = wave.min(1, keepdim=True).values, wave.max(1, keepdim=True).values
    wave = (wave - lo) / (hi - lo) * 2 - 1
    bins = torch.arange(1, lt + 1, dtype=torch.float32)[None, :]
    theta = 3 + 12 * torch.rand(n, 1, generator=g)
    pdf = bins * torch.exp(-bins / theta)
    isi = pdf / pdf.sum(1, keepdim=True) + (1e-3 * torch.randn(n, lt, generator=g)).abs()
    isi = torch.log1p(isi)
    labels = torch.randint(1, 5, (n,), generator=g)
    return wave.float().to(device), isi.float().to(device), labels.to(device)


class Pair:
    """wave + time models: two engines, each replaying its hipGraph(s) on its own HIP stream (the GPU overlaps the two
    models' kernels).  (A zipped two-model program on one stream was built in round 2 and measured slower — 98.3 k vs
    113.8 k samples/s: nothing hides the launch floors on a single stream — and removed in round 3)"""

    def __init__(self, device, world, lr=1e-3, lens=(50, 100), lockstep=False, fuse_bn=True, mfma_dtype="f32", reuse_ws=True,
                 model_type="unimodal", staged=True, rank=0, bucketed=False, act_dtype="f32"):
        self.device, self.world, self.lockstep = device, world, lockstep
        # staged: the synthetic tables are RESIDENT in each engine's workspace and every step's batch gather + eps draw is the first
        # launch of the step's graph (HP_OP_STAGE_BATCH): one graph replay per model-step, no torch kernel inside the timed region.
        # False (--no-staged, A/B): torch index_select / copy_ / normal_ in front of every step, as in rounds 1-2.
        self.staged = staged
        res = dict(resident_units=N_UNITS, dp_world=world, dp_rank=rank) if staged else {}
        # data parallel: the backward pass in two halves with the decoder-side gradient bucket all-reduced (communicator's stream) while the
        # encoder-side half runs on the model's own stream (hippie_amd.parallel.backward_allreduce)
        self.bucketed = bucketed
        res["bucketed_bwd"] = bucketed
        res["act_dtype"] = act_dtype          # "bf16": the backbones' activations stored as bfloat16 (only with mfma_dtype="bf16")
        self.only = None          # --only-model: step one of the two models (how much of the pair step is overlap?)
        self.multimodal = model_type == "multimodal"
        if self.multimodal:
            # MultiModalCVAE + MultiModalCVAETrainModule (hippie/model.py:350-533): ONE model with two encoder / decoder towers, one
            # optimisation step per batch; the script's multimodal trainer clips gradients (scripts/...:701)
            cfgs = [planner.ModelCfg(kind="multimodal", z_dim=Z_DIM, output_size=lens[0], output_size2=lens[1])]
            tcs = [planner.TrainCfg(lr=lr, weight_decay=0.01, beta=1.0, clip=1.0, fuse_bn=fuse_bn, mfma_dtype=mfma_dtype, reuse_workspace=reuse_ws, **res)]
        else:
            cfgs = [planner.ModelCfg(kind="unimodal", z_dim=Z_DIM, output_size=lens[0]), planner.ModelCfg(kind="unimodal", z_dim=Z_DIM, output_size=lens[1])]
            tcs = [planner.TrainCfg(lr=lr, weight_decay=0.01, beta=1.0, clip=0.0, fuse_bn=fuse_bn, mfma_dtype=mfma_dtype, reuse_workspace=reuse_ws, **res),
                   planner.TrainCfg(lr=lr, weight_decay=0.01, beta=1.0, clip=1.0, fuse_bn=fuse_bn, mfma_dtype=mfma_dtype, reuse_workspace=reuse_ws, **res)]
        self.eng = [Engine(c, BATCH, t, device=device) for c, t in zip(cfgs, tcs)]
        self.streams = [torch.cuda.Stream(device=device) for _ in self.eng]
        self.stream_pair = {}
        self.stream_priority = True      # --no-stream-priority (A/B): both model streams at normal priority
        self.run_ahead = 2               # --run-ahead: how many steps the host may queue ahead of the GPU (0 = unbounded)
        self.groups, self.comm_streams, self.comm_report = None, None, None
        if world > 1 or os.environ.get("HIPPIE_FORCE_DIST"):
            self.use_world_group()
        self.init_params()

    def pick_streams(self):
        """Which two HIP streams overlap is a lottery of hardware-queue assignment (hippie_amd/streams.py: 4.4 / 5.6 / 8 ms per pair-step
        for concurrent / same-queue / time-slicing pairs): measured on the evaluation-forward graphs, after RCCL has made its streams."""
        from hippie_amd.streams import pick_concurrent_streams
        self.stream_pair.clear()
        self.streams = pick_concurrent_streams(self.eng, self.device, report=self.stream_pair, refresh=True, prioritise_longer=self.stream_priority)

    def use_world_group(self):
        """gradient mean-all-reduce between bwd and opt over the default process group.  One communicator: the two models'
        all-reduces are issued in the same order on every rank and serialise on its stream (the wave model's backward
        finishes first anyway); each one overlaps the OTHER model's kernels, which run on their own stream."""
        import torch.distributed as dist
        self.groups = [dist.group.WORLD for _ in self.eng]
        self.comm_streams = [None for _ in self.eng]
        if self.bucketed:
            # the two-bucket form: a communicator and a communication stream PER MODEL — the models' collectives never queue behind each
            # other, and every side stream is probed not to share a hardware queue with a model stream (pick_comm_streams, after the
            # model streams are chosen)
            self.groups = [dist.new_group(list(range(dist.get_world_size()))) for _ in self.eng]
            for g in self.groups:
                dist.all_reduce(torch.zeros(1, device=self.device), group=g)

    def pick_comm_streams(self):
        from hippie_amd.streams import pick_side_stream
        busy = list(self.streams)
        self.comm_report = []
        for k in range(len(self.eng)):
            rep = {}
            self.comm_streams[k] = pick_side_stream(busy, self.device, report=rep)
            busy.append(self.comm_streams[k])
            self.comm_report.append(rep)

    def load_tables(self, data, perm):
        """staged mode: the tables, the shuffle and a noise seed into every engine's workspace (once, before the timed region)"""
        for k, e in enumerate(self.eng):
            if self.multimodal:
                e.load_dataset(data[0], data[2], x2=data[1], perm=perm, seed=1234)
            else:
                e.load_dataset(data[k], data[2], perm=perm, seed=1234 + k)

    def stage(self, k, e, data, idx):
        """gather the batch by index from the HBM-resident tables into the engine's input slots; draw eps"""
        if self.multimodal:
            e.io("x").copy_(data[0].index_select(0, idx).view(BATCH, 1, -1), non_blocking=True)
            e.io("x2").copy_(data[1].index_select(0, idx).view(BATCH, 1, -1), non_blocking=True)
        else:
            e.io("x").copy_(data[k].index_select(0, idx).view(BATCH, 1, -1), non_blocking=True)
        e.io("src").copy_(data[2].index_select(0, idx), non_blocking=True)
        e.io("eps").normal_()

    def init_params(self):
        """the reference constructors' initialisation (hippie_amd.model.reference_init_state: kaiming-uniform(a=sqrt 5) weights,
        U(+-1/sqrt(fan_in)) biases, BatchNorm 1 / 0, Embedding N(0,1)) from a fixed seed"""
        from hippie_amd.model import reference_init_state
        g = torch.Generator(device="cpu").manual_seed(42)
        for e in self.eng:
            e.load_state_dict(reference_init_state(e.cfg, g), strict=False)

    def fork(self):
        """The model streams start after whatever is queued on the current stream."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def step(self, data, idx, use_graph=True):
        """One optimisation step of each model.  The two models' step sequences are independent of each other
        (the reference trains them one after the other), so unless --lockstep is given their streams are joined
        only at the ends of a run (fork()/join()), not per step: the shorter wave step does not wait for the
        time step."""
        if self.lockstep:
            self.fork()
        for k, (e, s) in enumerate(zip(self.eng, self.streams)):
            if self.only is not None and k != self.only:
                continue
            with torch.cuda.stream(s):
                if self.staged:
                    if self.groups is None:
                        e.train_step_staged(use_graph)   # gather + eps + fwd + bwd + opt: ONE graph replay, nothing else on the stream
                        continue
                    e.staged_forward(use_graph)
                else:
                    self.stage(k, e, data, idx)
                    if self.groups is None:
                        e.train_step(use_graph)          # one process: nothing sits between bwd and opt -> one graph per step
                        continue
                    e.forward(True, use_graph)
                parallel.backward_allreduce(e, self.groups[k], use_graph, comm_stream=self.comm_streams[k])
                e.optimizer_step(use_graph)
        if self.lockstep:
            self.join()

    def run(self, n, data, batch_idx, first, use_graph=True):
        """n steps with the host at most `run_ahead` steps ahead of the slower model stream (0: unbounded — everything is queued within
        a few milliseconds).  A bounded run-ahead keeps the two free-running models within a couple of steps of each other and the
        hardware queues shallow: tools/micro/runahead_probe.py, 4.32-4.33 ms per pair-step at a depth of 2 against 4.35-4.38 unbounded
        (4.40 against 4.43-4.45 in a 20-step run); a depth of 1 stalls the pipeline (4.52)."""
        D = self.run_ahead
        ring = [[None] * max(D, 1) for _ in self.streams]
        for i in range(n):
            if D and i >= D:
                for k in range(len(self.streams)):
                    ring[k][i % D].synchronize()
            self.step(data, batch_idx(first + i), use_graph)
            if D:
                for k, s in enumerate(self.streams):
                    ev = torch.cuda.Event()
                    ev.record(s)
                    ring[k][i % D] = ev


def _nbuf(r, slots):
    return sum(1 for k in slots if int(r["buf"][k]) != P.NULL)


def op_bytes(r):
    """Algorithmic HBM bytes of one HBM/latency-bound op record (every operand read once, every result written once)."""
    opc, I = int(r["op"]), r["i"]
    es = 2 if int(r["flags"]) & P.FLAG_ACT_BF16 else 4          # bytes per stored activation element
    if opc == P.BN_APPLY:                 # raw (+ residual tensor / second raw) -> out
        return es * int(I[0]) * int(I[1]) * (2 + (1 if int(I[2]) else 0))
    if opc == P.BN_BWD_REDUCE:            # g1 (+g2) (+act) + raw (+raw2) -> g
        return es * int(I[0]) * int(I[1]) * (1 + _nbuf(r, (0, 1, 2, 4, 7)))
    if opc == P.BN_BWD_APPLY:             # g, raw -> dr
        return es * int(I[0]) * int(I[1]) * 3
    if opc == P.ADAMW:
        return 28 * int(I[0])
    if opc == P.GRADNORM:
        return 4 * int(I[0])
    if opc == P.REPARAM_KL_FWD:
        return 16 * int(I[0]) * int(I[1])
    if opc == P.REPARAM_KL_BWD:
        return 20 * int(I[0]) * int(I[1])
    if opc in (P.LINEAR_FWD, P.LINEAR_BWD_X, P.LINEAR_BWD_W):      # the skinny GEMMs of the heads: X + W + Y
        M, N, K = int(I[0]), int(I[1]), int(I[2])
        return 4 * (M * K + N * K + M * N)
    if opc == P.ZERO:
        return int(I[0]) + (int(I[1]) << 32)
    return None


def profile_ops(pair, data, idx, reps=5):
    """Per-launch HIP-event timing (hp_program_profile: events on the stream the kernels are launched on) of every op
    of one pair-step, in an untimed eager pass over the same programs.  Returns one dict per LAUNCH:
    {model, seg, kind, note, us, flop (conv / wgrad: 2*M*N*K*taps over the launch's member records), bytes, members}."""
    rows = []
    progs = []
    for k, e in enumerate(pair.eng):
        pair.stage(k, e, data, idx)
        progs.append((k, e.ops, e.plan.ops.segments, e.plan.ops.notes,
                      (lambda seg, e=e: e.prog.profile(*e.plan.ops.segments[seg], torch.cuda.current_stream().cuda_stream))))
    for model, ops, segments, notes, prof in progs:
        for seg in ("stage", "fwd_train", "bwd", "opt"):
            if seg not in segments:
                continue
            first, count = segments[seg]
            prof(seg)                                   # untimed: first eager launches load code objects / build tables
            acc = np.median(np.stack([prof(seg) for _ in range(reps)]), axis=0).astype(np.float64)      # median: robust against a stray slow launch
            for j in range(count):
                r = ops[first + j]
                opc = int(r["op"])
                if int(r["flags"]) & P.FLAG_MEMBER or opc == P.STATS_SYNC:
                    continue                                  # executed (and timed) by its PAIR / WGRAD_GROUP launch
                members, kind = [r], P.OP_NAMES[opc]
                if opc == P.PAIR:
                    members = [ops[int(r["i"][0])], ops[int(r["i"][1])]]
                    kind = "PAIR:" + P.OP_NAMES[int(members[0]["op"])]
                elif opc == P.WGRAD_GROUP:
                    members = list(ops[int(r["i"][0]): int(r["i"][0]) + int(r["i"][1])])
                mop = int(members[0]["op"])
                flop = None
                if mop in (P.CONV_TAPS, P.WGRAD_TAPS):
                    flop = sum(2.0 * int(m["i"][0]) * int(m["i"][1]) * int(m["i"][2]) * int(m["i"][9]) for m in members)
                nbytes = [op_bytes(m) for m in members]
                rows.append(dict(model=model, seg=seg, kind=kind, base=P.OP_NAMES[mop], note=notes[first + j], us=float(acc[j]) * 1e3, flop=flop,
                                 bytes=sum(nbytes) if all(v is not None for v in nbytes) else None, members=members))
    return rows


def conv_back_to_back(pair, reps=20):
    """Secondary figure for the conv kernel: every conv launch of one pair-step (CONV_TAPS records and PAIRs of them)
    repeated `reps` times back to back in a captured graph on the engine's own arenas, HIP events around the replay.
    The eager per-launch pass above brackets every launch with two event records (~4 us of floor per launch); this one
    leaves only the 1.6 us dependent-launch floor of the graph.  -> (sum of per-launch us over the step, launches)"""
    from hippie_amd.program import DeviceProgram
    total_us, launches = 0.0, 0
    stream = torch.cuda.current_stream().cuda_stream
    for e in pair.eng:
        arenas = [e.ws, e.params, e.grads, e.bufs, e.m, e.v]
        bases, sizes = [a.data_ptr() for a in arenas], [a.numel() * a.element_size() for a in arenas]
        ops, segs = e.ops, e.plan.ops.segments
        for seg in ("fwd_train", "bwd"):
            first, count = segs[seg]
            for g in range(first, first + count):
                r = ops[g]
                opc = int(r["op"])
                if int(r["flags"]) & P.FLAG_MEMBER:
                    continue
                if opc == P.CONV_TAPS:
                    unit = [r]
                elif opc == P.PAIR and int(ops[int(r["i"][0])]["op"]) == P.CONV_TAPS:
                    unit = [ops[int(r["i"][0])], ops[int(r["i"][1])], r]
                else:
                    continue
                recs = []
                for rep in range(reps):
                    for u in unit:
                        u = u.copy()
                        if int(u["op"]) == P.PAIR:
                            u["i"][0], u["i"][1] = rep * 3, rep * 3 + 1
                        recs.append(u)
                prog = DeviceProgram(np.array(recs, dtype=P.OP_DTYPE), bases, sizes)
                gid = prog.capture(0, len(recs))
                prog.replay(gid, stream)
                best = 1e30
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    prog.replay(gid, stream)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
                prog.close()
                total_us += best
                launches += 1
    return total_us, launches


def summarize(rows):
    """-> (conv roofline numbers, per-kind table, HBM GB/s of the bandwidth kernels, encoder-forward MFMA fraction)"""
    conv = [r for r in rows if r["base"] == "CONV_TAPS"]
    conv_us, conv_flop = sum(r["us"] for r in conv), sum(r["flop"] for r in conv)
    enc_fwd = [r for r in conv if r["seg"] == "fwd_train" and "encoder" in r["note"]]
    ef_us, ef_flop = sum(r["us"] for r in enc_fwd), sum(r["flop"] for r in enc_fwd)
    # the whole encoder forward phase (stem .. pooled features): its convs' FLOPs over ALL its launches' time
    enc_phase_us = sum(r["us"] for r in rows if r["seg"] == "fwd_train" and "encoder" in r["note"] and "encoder_fc" not in r["note"])
    per_kind = {}
    for r in rows:
        d = per_kind.setdefault(r["kind"], [0.0, 0])
        d[0] += r["us"]
        d[1] += 1
    hbm = {}
    for name, bases in (("bn_apply", ("BN_APPLY",)), ("bn_bwd_reduce", ("BN_BWD_REDUCE",)), ("bn_bwd_apply", ("BN_BWD_APPLY",)),
                        ("adamw", ("ADAMW",)), ("gradnorm", ("GRADNORM",)), ("reparam_kl", ("REPARAM_KL_FWD", "REPARAM_KL_BWD")),
                        ("skinny_linear", ("LINEAR_FWD", "LINEAR_BWD_X", "LINEAR_BWD_W"))):
        sel = [r for r in rows if r["base"] in bases and r["bytes"] is not None]
        if sel:
            us = sum(r["us"] for r in sel)
            hbm[name] = {"gbps": sum(r["bytes"] for r in sel) / (us * 1e-6) / 1e9, "launches": len(sel), "avg_us": us / len(sel),
                         "frac_of_8TBps": sum(r["bytes"] for r in sel) / (us * 1e-6) / 8e12}
    wg = [r for r in rows if r["base"] == "WGRAD_TAPS"]
    wgrad = {"tflops": sum(r["flop"] for r in wg) / (sum(r["us"] for r in wg) * 1e-6) / 1e12, "launches": len(wg),
             "us_per_step": sum(r["us"] for r in wg)} if wg else None
    return dict(conv_us=conv_us, conv_flop=conv_flop, conv_launches=len(conv), enc_fwd_tflops=ef_flop / (ef_us * 1e-6) / 1e12,
                enc_fwd_phase_tflops=ef_flop / (enc_phase_us * 1e-6) / 1e12, per_kind=per_kind, hbm=hbm, wgrad=wgrad,
                total_us=sum(r["us"] for r in rows), launches=len(rows))


def csrc_digest():
    """sha256 over the kernel sources: a PMC capture is only quoted next to numbers measured on the same kernels"""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_mfma.hip", "ops_small.hip", "hp_common.h"):
        with open(os.path.join(ROOT, "hippie_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_from_pmc(args):
    """HBM bytes per conv launch from committed rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled as the
    gfx950 guide prescribes + WRITE_SIZE, KB -> bytes).  Only when the capture was taken on THESE kernels (source
    digest) at THIS shape; otherwise null — a stale constant next to fresh FLOP/s would be meaningless."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        meta = d.get("meta", {})
        same = (meta.get("csrc_digest") == csrc_digest() and
                [meta.get(k) for k in ("batch", "z_dim", "wave_len", "time_len")] ==
                [args.batch, args.z_dim, args.wave_len, args.time_len])
        return (d["hbm_bytes_per_launch"], {"file": os.path.relpath(PMC_SUMMARY, ROOT), "git_head": meta.get("git_head")}) if same else (None, None)
    except Exception:
        return None, None


class HbmLoader:
    """DataLoader(batch_size, shuffle=True) over HBM-resident tensors: ([B,1,L], labels[B]) batches, last one ragged."""

    def __init__(self, x, labels, batch, seed):
        self.x, self.labels, self.batch = x, labels, batch
        self.gen = torch.Generator(device="cpu").manual_seed(seed)

    def __len__(self):
        return -(-self.x.shape[0] // self.batch)

    def _batches(self, perm):
        for i in range(0, len(perm), self.batch):
            j = perm[i: i + self.batch]
            yield self.x.index_select(0, j).unsqueeze(1), self.labels.index_select(0, j)

    def __iter__(self):
        return self._batches(torch.randperm(self.x.shape[0], generator=self.gen).to(self.x.device))

    def frozen(self):
        """one pass with its shuffle drawn now (hippie_amd.trainer.fit_concurrently)"""
        perm = torch.randperm(self.x.shape[0], generator=self.gen).to(self.x.device)
        loader = self

        class _Pass:
            def __iter__(p):
                return loader._batches(perm)

            def __len__(p):
                return len(loader)
        return _Pass()


def trainer_rate(device, data, epochs=4):
    """Throughput of the REFERENCE-API path (what scripts/pretrain_pipeline.py runs): hippieUnimodalCVAE +
    hippieUnimodalEmbeddingModelCVAE driven by Trainer.fit over the synthetic pretrain pool at batch 512 — the wave model (no
    clipping) and the time model (clip 1.0) of scripts/train_model_with_multimodal.py:200-224, fitted CONCURRENTLY on two HIP
    streams by hippie_amd.trainer.fit_concurrently (same random draws and numbers as one after the other; round 2 ran them
    sequentially: 90 k samples/s).  A ragged last batch per epoch, per-epoch shuffling by index.
    samples/s = N * epochs / wall time of the concurrent fits."""
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    from hippie_amd.trainer import Trainer, fit_concurrently
    N = data[0].shape[0]
    jobs, warm = [], []
    for k, (L, clip) in enumerate(((data[0].shape[1], None), (data[1].shape[1], 1.0))):
        net = hippieUnimodalCVAE(z_dim=Z_DIM, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5, device=device)
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3, weight_decay=0.01)
        loader = HbmLoader(data[k], data[2], BATCH, seed=7 + k)
        warm.append((Trainer(max_epochs=1, gradient_clip_val=clip, enable_checkpointing=False, num_sanity_val_steps=0), mod, loader, None))
        jobs.append((Trainer(max_epochs=epochs, gradient_clip_val=clip, enable_checkpointing=False, num_sanity_val_steps=0), mod, loader, None))
    fit_concurrently(warm)          # warm-up epoch: lowers, captures the graphs, measures which stream pair overlaps (hippie_amd/streams.py)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit_concurrently(jobs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": N * epochs / dt, "unit": "samples/s", "epochs": epochs, "units_per_epoch": N, "wall_s": dt,
            "path": "hippie_amd.model.hippieUnimodalEmbeddingModelCVAE.training_step / optimizer.step via hippie_amd.trainer.Trainer.fit "
                    "(hipGraph replay, per-step losses kept on the device); wave and time fits concurrently on two streams "
                    "(hippie_amd.trainer.fit_concurrently), as scripts/pretrain_pipeline.py runs them"}


def embedding_rate(device, data, passes=3):
    """Throughput of the REFERENCE-API inference path (scripts/inference_from_trained_model.py:130-151 -> scripts/utils.py:75-101):
    hippie_amd.utils.get_embeddings over the whole synthetic pool at batch 512 — both modules in eval mode, `enc` of every batch
    (eval-mode BatchNorm folded into the conv epilogues, decoder skipped), row-standardised, wave | time concatenated, copied to the host.
    units/s = N * passes / wall time."""
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    from hippie_amd.utils import get_embeddings
    N = data[0].shape[0]
    mods, loaders = [], []
    for k in range(2):
        net = hippieUnimodalCVAE(z_dim=Z_DIM, output_size=data[k].shape[1], class_hidden_dim=5, num_sources=5, num_classes=5, device=device)
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3, weight_decay=0.01)
        mod.eval()
        mods.append(mod)
        x, labels = data[k], data[2]
        loaders.append([(x[i: i + BATCH].unsqueeze(1), labels[i: i + BATCH]) for i in range(0, N, BATCH)])
    get_embeddings(loaders[0], loaders[1], mods[0], mods[1])          # warm-up: lowers + captures the eval graphs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        ew, et, joint = get_embeddings(loaders[0], loaders[1], mods[0], mods[1])
    dt = time.perf_counter() - t0
    assert joint.shape == (N, 2 * Z_DIM) and np.isfinite(joint).all()
    return {"value": N * passes / dt, "unit": "units/s", "passes": passes, "units_per_pass": N, "batch": BATCH, "wall_s": dt,
            "path": "hippie_amd.utils.get_embeddings (wave + time modules in eval mode, encoder half only, row-standardised embeddings on the host)"}


def cpu_baseline(steps=20, warm=3):
    """The torch-CPU oracle (kind 'port': a restatement of the reference's PyTorch path, pinned to it
    by tests/golden) on this host: wave step + time step (clip 1.0) at batch 512; median of `steps` timed
    steps after `warm` warm-up steps."""
    from oracle import cvae_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    times = []
    for L, clip in ((50, None), (100, 1.0)):
        m = O.OracleModel("unimodal", Z_DIM, L)
        x, src, cls, eps = O.synth_inputs(BATCH, L, Z_DIM)
        for _ in range(warm):
            m.train_step((x, src, None), eps, lr=1e-6, clip=clip)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            m.train_step((x, src, None), eps, lr=1e-6, clip=clip)
            ts.append(time.perf_counter() - t0)
        times.append(float(np.median(ts)))
    # ... and with the reference's loader in front of every step ("also report the loader-inclusive figure"): per-item
    # float cast / log1p / F.interpolate / stack for both modalities from raw tables of the shipped widths (47 and 100 columns)
    from oracle import preproc
    rng = np.random.default_rng(0)
    raw = (rng.standard_normal((2048, 47)).astype(np.float64), np.abs(rng.standard_normal((2048, 100))).astype(np.float64))
    tl = []
    for rows, L, lg in ((raw[0], 50, False), (raw[1], 100, True)):
        ts = []
        for r in range(5):
            idx = rng.integers(0, 2048, BATCH)
            t0 = time.perf_counter()
            b = preproc.reference_style_batch(rows, L, lg, idx)
            ts.append(time.perf_counter() - t0)
        assert tuple(b.shape) == (BATCH, 1, L)
        tl.append(float(np.median(ts)))
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except Exception:
        pass
    return dict(value=BATCH / sum(times), unit="samples/s", cores=cores, kind="port", cpu_model=cpu_model, torch=torch.__version__,
                loader_inclusive_value=BATCH / (sum(times) + sum(tl)),
                loader_ms_per_batch={"wave": round(tl[0] * 1e3, 2), "time": round(tl[1] * 1e3, 2)},
                sample=f"median of {steps} steps (after {warm} warm-up) of wave (L=50) + time (L=100, clip 1.0) cVAE at batch 512, torch {torch.__version__} CPU, "
                       f"wave {times[0]*1e3:.0f} ms + time {times[1]*1e3:.0f} ms per step; loader_inclusive_value adds the reference's per-item loader "
                       f"(float cast, log1p, F.interpolate, stack: {tl[0]*1e3:.1f} + {tl[1]*1e3:.1f} ms per batch of 512, one process as num_workers=0)")


def spawn_ranks(n):
    """One child per GPU (rank r on device r), rendezvous on 127.0.0.1; returns the exit code for the parent."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    if any(codes):
        for p in procs:
            if p.poll() is None:
                p.kill()
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    lines = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    global BATCH, Z_DIM, N_UNITS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event pass (roofline fields null): for runs under rocprofv3 --pmc")
    ap.add_argument("--no-trainer", action="store_true", help="skip the secondary reference-API (Trainer.fit) throughput measurement")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-reuse-ws", action="store_true", help="every workspace tensor in memory of its own (A/B against the liveness-packed arena)")
    ap.add_argument("--only-model", type=int, choices=(0, 1), default=None, help="diagnostic: step only the wave (0) or the time (1) model; the line is then NOT the headline metric")
    ap.add_argument("--no-pick-streams", action="store_true", help="A/B: take the first two streams torch hands out instead of measuring which pair overlaps")
    ap.add_argument("--run-ahead", type=int, default=2, help="steps the host may queue ahead of the GPU (0 = unbounded, the A/B)")
    ap.add_argument("--no-stream-priority", action="store_true", help="A/B: no high-priority stream for the longer chain (the time model)")
    ap.add_argument("--lockstep", action="store_true", help="join the two model streams after every step (default: only at the ends of the run)")
    ap.add_argument("--per-op", action="store_true", help="print the per-op time table to stderr")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the headline: the reference's arithmetic) | bf16: BASELINE config 2's reduced-precision mode "
                         "(bf16 MFMA operands, fp32 accumulation / statistics / master weights): a separately labelled line, never the headline")
    ap.add_argument("--matrix-path", choices=["bf16x3", "f32"], default="bf16x3",
                    help="with --dtype f32: which matrix cores carry the fp32 arithmetic.  bf16x3 (default, the product default): every fp32 operand "
                         "split exactly into three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation — error vs "
                         "fp64 at or below the fp32 matrix path's (tests/test_gpu_split.py), same parity suite.  f32: v_mfma_f32_32x32x2_f32 (rounds 1-3)")
    ap.add_argument("--bf16-f32-storage", action="store_true",
                    help="A/B with --dtype bf16: keep the activation tensors in fp32 and round only the matrix operands (round 3's form).  Default with "
                         "--dtype bf16: the backbones' activations and their gradients are also STORED as bfloat16 (TrainCfg.act_dtype: faster at every "
                         "batch size measured and a third less workspace; profiles/r04_bf16_storage.txt)")
    ap.add_argument("--no-fuse-bn", action="store_true", help="A/B: one launch per BatchNorm pass instead of the fused loaders / epilogues")
    # non-default shapes (BASELINE configs[2]: --batch 4096 --z-dim 32 --wave-len 256 --time-len 32); the headline
    # metric is always quoted on the defaults
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--z-dim", type=int, default=10)
    ap.add_argument("--wave-len", type=int, default=50)
    ap.add_argument("--time-len", type=int, default=100)
    ap.add_argument("--units", type=int, default=15631)
    ap.add_argument("--model-type", choices=["unimodal", "multimodal"], default="unimodal",
                    help="unimodal (default, the headline): the wave + time cVAE pair.  multimodal: ONE MultiModalCVAE step per batch "
                         "(BASELINE configs[4]'s per-rank shape: --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32): "
                         "a NON-DEFAULT, separately labelled line")
    ap.add_argument("--no-staged", action="store_true", help="A/B: stage every batch with torch kernels (index_select / copy_ / normal_) instead of the "
                    "in-graph HP_OP_STAGE_BATCH over workspace-resident tables")
    ap.add_argument("--bucketed-bwd", action="store_true",
                    help="A/B (N > 1 and the 1-rank probe): the backward pass in two halves with the decoder-side gradient bucket all-reduced on a side "
                         "stream under the encoder-side half, instead of one all-reduce of the whole arena after the pass (the default: measured "
                         "faster at one rank on this runtime, DESIGN.md section 6)")
    ap.add_argument("--no-dp-probe", action="store_true", help="skip the 1-rank RCCL data-parallel overhead probe (N=1 only)")
    args = ap.parse_args()
    args.mm = "bf16" if args.dtype == "bf16" else args.matrix_path      # planner.TrainCfg.mfma_dtype of this run
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one child process per GPU BEFORE anything in this
        # process touches the GPU (a process that has initialised HIP must never exec / be replaced), forward rank
        # 0's JSON line, fail if any rank fails.
        raise SystemExit(spawn_ranks(args.gpus))
    # stdout carries exactly ONE line, the JSON: native libraries print there too (RCCL's version banner at
    # communicator creation), so fd 1 is pointed at stderr for the run and the JSON goes to the saved descriptor
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    BATCH, Z_DIM, N_UNITS = args.batch, args.z_dim, max(args.units, args.batch * 2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HIPPIE_SINGLE_DEVICE"):                       # rehearsal: every rank on GPU 0
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = bool(os.environ.get("HIPPIE_FORCE_DIST"))          # experiment: 1-rank RCCL, all-reduce still issued
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HIPPIE_DIST_BACKEND", "nccl")      # nccl = RCCL on ROCm; gloo only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    n_ranks_seen = dist.get_world_size() if (world > 1 or force_dist) else 1      # what the communicator says, not the flag
    dist_backend = dist.get_backend() if (world > 1 or force_dist) else None
    data = synth_dataset(N_UNITS, device, lw=args.wave_len, lt=args.time_len)
    pair = Pair(device, world, lens=(args.wave_len, args.time_len), lockstep=args.lockstep,
                fuse_bn=not args.no_fuse_bn, mfma_dtype=args.mm, reuse_ws=not args.no_reuse_ws, model_type=args.model_type,
                staged=not args.no_staged, rank=rank, bucketed=(world > 1 or force_dist) and args.bucketed_bwd,
                act_dtype="bf16" if (args.dtype == "bf16" and not args.bf16_f32_storage) else "f32")
    pair.only = args.only_model
    pair.stream_priority = not args.no_stream_priority
    pair.run_ahead = max(0, args.run_ahead)
    steps_per_epoch = N_UNITS // (BATCH * world)
    g = torch.Generator(device="cpu").manual_seed(1234)
    perm = torch.randperm(N_UNITS, generator=g).to(device)

    def batch_idx(i):
        j = (i % steps_per_epoch) * world + rank          # DistributedSampler-style interleave
        return perm[j * BATCH:(j + 1) * BATCH]

    use_graph = not args.no_graph
    if pair.staged:
        pair.load_tables(data, perm)
    if world > 1 or force_dist:
        dist.all_reduce(torch.zeros(1, device=device))       # the communicator's own stream exists before the model streams are chosen
    if not args.no_pick_streams and pair.only is None:
        pair.pick_streams()
    if pair.groups is not None and pair.bucketed:
        pair.pick_comm_streams()
    stream_pair = dict(pair.stream_pair)
    pair.fork()
    pair.run(args.warmup, data, batch_idx, 0, use_graph)
    pair.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pair.fork()
    pair.run(args.steps, data, batch_idx, args.warmup, use_graph)
    pair.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = [e.scalars()[0] for e in pair.eng]

    def timed(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        pair.fork()
        pair.run(n, data, batch_idx, args.warmup, use_graph)
        pair.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    dp_probe = None
    if world == 1 and not force_dist and not args.no_dp_probe and not args.no_profile and pair.only is None:
        # What the data-parallel structure costs BEFORE any wire time: the same step with a 1-rank RCCL communicator in place —
        # three graphs per model-step instead of one, ncclAllReduce(AVG) of the gradient arena between bwd and opt on the
        # communicator's stream (a copy onto itself at one rank).  N > 1 adds only the transfer time on top of this.
        main_pair = pair
        try:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ["HIPPIE_FORCE_DIST"] = "1"
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            if args.bucketed_bwd:
                # the N > 1 lowering (TrainCfg.bucketed_bwd: two backward halves, two gradient buckets) in engines of its own
                pair = Pair(device, 1, lens=(args.wave_len, args.time_len), lockstep=args.lockstep, fuse_bn=not args.no_fuse_bn, mfma_dtype=args.mm,
                            reuse_ws=not args.no_reuse_ws, model_type=args.model_type, staged=not args.no_staged, rank=0, bucketed=True,
                            act_dtype="bf16" if (args.dtype == "bf16" and not args.bf16_f32_storage) else "f32")
                pair.stream_priority, pair.run_ahead = main_pair.stream_priority, main_pair.run_ahead
                if pair.staged:
                    pair.load_tables(data, perm)
            pair.use_world_group()
            dist.all_reduce(torch.zeros(1, device=device))
            if not args.no_pick_streams:
                pair.pick_streams()
            if pair.bucketed:
                pair.pick_comm_streams()
            timed(args.warmup)
            dt_dp = timed(args.steps)
            dp_probe = {"ms_per_step": dt_dp / args.steps * 1e3, "value": BATCH * args.steps / dt_dp, "ratio_to_value": dt / dt_dp,
                        "bucketed_bwd": bool(pair.bucketed), "comm_streams": pair.comm_report,
                        "how": "the N > 1 step at one rank: stage+forward | bwd_dec | async RCCL all-reduce of the decoder-side gradient bucket on the "
                               "communicator's stream | bwd_enc on the model's own stream | all-reduce of the encoder-side bucket | wait | opt "
                               "(4 graph replays per model-step); N > 1 adds the xGMI transfer time of the two buckets, the first under bwd_enc"
                               if pair.bucketed else
                               "same step with a 1-rank RCCL (nccl) all-reduce of each model's gradient arena between bwd and opt (3 graph replays per model-step)"}
            dist.destroy_process_group()
        except Exception as ex:                 # never lose the line over a secondary figure
            dp_probe = {"error": repr(ex)[:200]}
        finally:
            os.environ.pop("HIPPIE_FORCE_DIST", None)
            pair = main_pair              # (the probe's own engines are dropped)
            pair.groups = None

    METRIC = "pretrain samples/sec (waveform+time cVAE, batch 512) at 1/2/4/8 MI355X"
    if rank == 0 and args.no_profile:
        out = {"metric": METRIC, "value": BATCH * world * args.steps / dt,
               "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "note": "--no-profile run (counter collection): no roofline fields"}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    elif rank == 0:
        rows = profile_ops(pair, data, batch_idx(0))
        sm = summarize(rows)
        conv_ms, conv_flop, launches = sm["conv_us"] * 1e-3, sm["conv_flop"], sm["conv_launches"]
        achieved = conv_flop / (conv_ms * 1e-3) / 1e12
        if args.per_op:
            for name, (us, cnt) in sorted(sm["per_kind"].items(), key=lambda kv: -kv[1][0]):
                print(f"{name:18s} {cnt:4d} launches {us / 1e3:8.3f} ms {100 * us / sm['total_us']:5.1f} %", file=sys.stderr)
            print(f"{'TOTAL (eager, serial)':18s} {sm['launches']:4d} launches {sm['total_us'] / 1e3:8.3f} ms", file=sys.stderr)
            for r in rows:
                m0 = r["members"][0]
                extra = ""
                if r["flop"]:
                    extra = "M=%6d N=%4d K=%4d taps=%d n=%d %6.1f TF" % (int(m0["i"][0]), int(m0["i"][1]), int(m0["i"][2]), int(m0["i"][9]),
                                                                          len(r["members"]), r["flop"] / (r["us"] * 1e-6) / 1e12)
                elif r["bytes"]:
                    extra = "%7.2f MB %7.1f GB/s" % (r["bytes"] / 1e6, r["bytes"] / (r["us"] * 1e-6) / 1e9)
                print("m%s %-9s %-16s %-70s %8.1f us  %s" % (r["model"], r["seg"], r["kind"], r["note"][:70], r["us"], extra), file=sys.stderr)
        traffic, traffic_src = traffic_from_pmc(args)
        try:
            b2b_us, b2b_n = conv_back_to_back(pair)
            b2b = {"avg_launch_us": b2b_us / b2b_n, "launches_per_step": b2b_n, "achieved": conv_flop / (b2b_us * 1e-6) / 1e12,
                   "frac": conv_flop / (b2b_us * 1e-6) / 1e12 / PEAK_TFLOPS[args.mm],
                   "how": "every conv launch repeated 20x back to back in a captured graph (HIP events around the replay): without the ~4 us "
                          "per-launch event floor of the eager pass behind `frac`; informational"}
        except Exception as ex:          # never lose the line over the secondary figure
            b2b = {"error": repr(ex)[:200]}
        out = {
            "metric": METRIC,
            "value": BATCH * world * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            # how the fp32 products reach the matrix cores (hippie_amd.planner.TrainCfg.mfma_dtype): "bf16x3" = exact three-term bf16 split of every
            # fp32 operand, six bf16 MFMA products per fp32 product, fp32 accumulation (fp32-level error: tests/test_gpu_split.py); "f32" = fp32 MFMA
            "matrix_path": args.mm,
            **({"activation_storage": "fp32" if args.bf16_f32_storage else "bf16"} if args.dtype == "bf16" else {}),
            "config": {"workload": ("REDUCED-PRECISION MODE (bf16 MFMA operands, fp32 accumulate; NOT the headline; tolerance: tests/test_gpu_bf16.py) — " if args.dtype == "bf16" else "") +
                                   ("BASELINE configs[1] shape: cellexplorer-celltype pretrain pool (15631 synthetic units), "
                                    "wave cVAE L=50 + time cVAE L=100 (clip 1.0), z_dim=10, per-GPU batch 512, AdamW lr 1e-3, "
                                    "fp32 arithmetic (the reference's arithmetic type; the config's bf16 wording is a separate, labelled mode) " +
                                    ("on the bf16 matrix cores through an exact three-term operand split (six products per fp32 product, fp32 accumulation)" if args.mm == "bf16x3" else "on f32 MFMA"))
                       if (args.batch, args.z_dim, args.wave_len, args.time_len, args.model_type) == (512, 10, 50, 100, "unimodal") else
                       (f"NON-DEFAULT (NOT the headline): ONE MultiModalCVAE (hippie/model.py:350-432) step per batch of {args.batch} units — wave L={args.wave_len} + "
                        f"time L={args.time_len} towers, z_dim={args.z_dim}, clip 1.0, {N_UNITS} synthetic units; BASELINE configs[4]'s per-rank shape "
                        "when --batch 8192 --z-dim 64 --wave-len 256 --time-len 32" if args.model_type == "multimodal" else
                        f"NON-DEFAULT shape: wave L={args.wave_len} + time L={args.time_len}, z_dim={args.z_dim}, batch {args.batch}, {N_UNITS} synthetic units"),
                       "model_type": args.model_type,
                       "global_batch": BATCH * world, "parallelism": f"dp{world}", "dist_backend": dist_backend, "hip_graph": use_graph, "staged_in_graph": pair.staged, "stream_pair": stream_pair or None, "run_ahead": pair.run_ahead, "fused_batchnorm": not args.no_fuse_bn, "workspace_mb": [round(e.plan.ws_bytes / 1e6, 1) for e in pair.eng], "workspace_unpacked_mb": [round((e.plan.ws_unpacked or e.plan.ws_bytes) / 1e6, 1) for e in pair.eng], "only_model": pair.only, "lockstep": pair.lockstep,
                       "final_loss_wave": loss[0], "final_loss_time": loss[-1],
                       # every HIPPIE_* variable of this run (measurement knobs act only under HIPPIE_DEBUG_KNOBS=1): {} = the product defaults
                       "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("HIPPIE_")}},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.mm], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS[args.mm], "traffic": traffic, "traffic_source": traffic_src,
                         # `achieved` counts ALGORITHMIC fp32 flops (2*M*N*K*taps); the bf16x3 path spends six bf16 MFMA products per fp32 product, so its
                         # peak is the dense bf16 peak / 6; the same figure against the fp32 matrix cores' peak, for comparison with rounds 1-3:
                         "peak_note": {"bf16x3": "2500 TFLOP/s dense bf16 / 6 products per fp32 product", "f32": "v_mfma_f32_32x32x2_f32 dense", "bf16": "dense bf16"}[args.mm],
                         "frac_of_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                         "kernel": "conv_taps_kernel + conv_taps_pair_kernel (same body: forward conv and input-gradient, " +
                                   {"bf16x3": "v_mfma_f32_32x32x16_bf16 x 6 on three-term operands", "f32": "f32 MFMA 32x32x2", "bf16": "v_mfma_f32_32x32x16_bf16"}[args.mm] +
                                   "; BatchNorm input transform in the loader / BatchNorm-backward reduction in the epilogue where fused)",
                         "launches_per_step": launches, "avg_launch_us": conv_ms * 1e3 / launches,
                         # ALGORITHMIC: sum over the launches' op records of 2*M*N*K*taps (no masked / zero-page taps exist any more)
                         "algorithmic_gflop_per_step": conv_flop / 1e9,
                         # north_star's ">= 40 % MFMA utilisation on the encoder forward": encoder forward convs' FLOPs over (a) those
                         # launches' time, (b) the whole encoder-forward phase incl. its BatchNorm / stem / pool launches
                         "encoder_forward": {"conv_tflops": sm["enc_fwd_tflops"], "conv_frac": sm["enc_fwd_tflops"] / PEAK_TFLOPS[args.mm],
                                             "phase_tflops": sm["enc_fwd_phase_tflops"], "phase_frac": sm["enc_fwd_phase_tflops"] / PEAK_TFLOPS[args.mm],
                                             # the target read on the WHOLE phase (conv launches alone: conv_frac); at batch 512 a layer has 200-448 tiles
                                             # for 256 CUs and the phase is bound by launch floors, not the K-loop: not met there
                                             "target": 0.40, "target_met": bool(sm["enc_fwd_phase_tflops"] / PEAK_TFLOPS[args.mm] >= 0.40),
                                             "target_met_on_conv_launches": bool(sm["enc_fwd_tflops"] / PEAK_TFLOPS[args.mm] >= 0.40),
                                             # the same two figures against the fp32 matrix cores' peak (what rounds 1-3 quoted; the three-term path's own peak is 2.65x that)
                                             "conv_frac_of_f32_mfma_peak": sm["enc_fwd_tflops"] / PEAK_F32_MFMA_TFLOPS,
                                             "phase_frac_of_f32_mfma_peak": sm["enc_fwd_phase_tflops"] / PEAK_F32_MFMA_TFLOPS},
                         "back_to_back": b2b,
                         "wgrad_group_kernel": sm["wgrad"],
                         # achieved HBM GB/s (algorithmic bytes / HIP-event time) of the bandwidth- and latency-bound kernels
                         "hbm_gbps": sm["hbm"],
                         "launches_per_pair_step": sm["launches"], "eager_serial_ms_per_pair_step": sm["total_us"] / 1e3,
                         # whole-step view: samples/s x 3 x forward FLOPs per unit, all kernels and gaps included
                         "whole_step_tflops_per_gpu": BATCH * args.steps / dt * 3.0 * sum(e.plan.flops_fwd for e in pair.eng) / BATCH / 1e12},
        }
        out["dp_overhead_1rank"] = dp_probe
        if world == 1 and not args.no_trainer and args.model_type == "unimodal":
            tr = trainer_rate(device, data) if args.dtype == "f32" else None
            out["trainer_samples_per_s"] = tr["value"] if tr else None
            out["trainer_path"] = tr
        if world == 1 and not args.no_trainer and args.model_type == "unimodal" and args.dtype == "f32":
            try:
                out["inference_path"] = embedding_rate(device, data)
            except Exception as ex:             # never lose the line over a secondary figure
                out["inference_path"] = {"error": repr(ex)[:200]}
        if not args.no_cpu_baseline and world == 1 and args.model_type == "unimodal":       # reported at N=1 only (the other ranks would sit in the barrier)
            out["cpu_baseline"] = cpu_baseline()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
