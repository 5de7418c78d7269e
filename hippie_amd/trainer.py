"""A minimal fit loop standing in for pytorch_lightning.Trainer (absent here) with the behaviour the
reference's scripts rely on (scripts/train_model_with_multimodal.py:186-230): automatic optimisation
(training_step -> backward -> [clip] -> optimizer.step), two sanity validation batches, a validation
pass per epoch, top-1 `val_loss` checkpointing (ModelCheckpoint(monitor="val_loss", mode="min")),
EarlyStopping(patience), and `.ckpt` files holding {"state_dict", "optimizer_states", "epoch"}.

Data parallel.  The reference passes no `devices` / `strategy` (scripts/...:200-207), so on a multi-GPU host Lightning
picks DDP by itself: one process per GPU, every DataLoader's sampler replaced by a DistributedSampler (per-rank batches
of `batch_size`), parameters broadcast from rank 0, gradients mean-all-reduced between backward and optimizer.step,
BatchNorm statistics local unless `sync_batchnorm=True`, checkpoints and logs written by rank 0.  `Trainer(strategy=
"ddp")` — or simply running under an initialised torch.distributed with more than one rank (torchrun) — gives the same
here, with RCCL ("nccl") as the collective backend; precision="bf16" (`bf16-mixed`) selects the bf16-MFMA lowering."""
from __future__ import annotations

import json
import os
import time

import torch


class Trainer:
    def __init__(self, max_epochs=1, gradient_clip_val=None, default_root_dir="checkpoints", patience=30,
                 monitor="val_loss", logger_path=None, num_sanity_val_steps=2, device=None, enable_checkpointing=True,
                 sync_every_step=False, deterministic=False, strategy="auto", devices="auto", sync_batchnorm=False,
                 precision="32", process_group=None, seed=0, run_ahead=2):
        """sync_every_step: True reproduces the reference's per-step `loss.item()` host sync (hippie/model.py:114) and
        checks labels on the host every step; False (default) keeps the step asynchronous — hipGraph replays, per-step
        losses kept on the device and averaged at the epoch end, label range errors raised at the epoch end.
        run_ahead: in the asynchronous mode, how many steps the host may queue ahead of the GPU (0 = unbounded).  Two keeps the
        hardware queue shallow and concurrent fits (fit_concurrently) within a couple of steps of each other — measured 1 % faster than
        queueing a whole epoch at once (bench.py Pair.run, tools/micro/runahead_probe.py)."""
        self.sync_every_step = sync_every_step
        self.run_ahead = max(0, int(run_ahead))
        self.deterministic = deterministic      # Lightning's flag: bit-reproducible runs (ordered weight-gradient sums, no fp32 atomics)
        self.max_epochs, self.gradient_clip_val = max_epochs, gradient_clip_val
        self.root, self.patience, self.monitor = default_root_dir, patience, monitor
        self.logger_path = logger_path
        self.num_sanity_val_steps = num_sanity_val_steps
        self.device = torch.device(device) if device is not None else None
        self.enable_checkpointing = enable_checkpointing
        self.best_model_path, self.best_score = "", float("inf")
        self.current_epoch, self.global_step = 0, 0
        self.history = []
        # ---- data parallel (Lightning's DDP strategy) ----
        if strategy not in ("auto", "ddp", None, "single_device"):
            raise ValueError(f"strategy {strategy!r}: only 'auto', 'ddp' and 'single_device' exist here (the model is 8 M parameters: nothing to shard but the batch)")
        if str(precision) not in ("32", "32-true", "bf16", "bf16-mixed"):
            raise ValueError(f"precision {precision!r}: '32' (the reference's arithmetic) or 'bf16' / 'bf16-mixed'")
        self.precision = "bf16" if str(precision).startswith("bf16") else "f32"
        self.strategy, self.devices, self.sync_batchnorm = strategy, devices, sync_batchnorm
        self.process_group, self.seed = process_group, seed
        self.world_size, self.global_rank = 1, 0

    # ---- data parallel plumbing ---------------------------------------------------------------------------------------
    def _setup_distributed(self):
        """world size / rank of this fit: >1 only when torch.distributed is initialised with more than one rank (the
        launcher's job — torchrun, or bench.py's own spawn: one process per GPU) and the strategy allows it."""
        import torch.distributed as dist
        self.world_size, self.global_rank = 1, 0
        if self.strategy in ("single_device", None) or not (dist.is_available() and dist.is_initialized()):
            if self.strategy == "ddp" and not isinstance(self.devices, str) and int(self.devices) > 1:
                raise RuntimeError("Trainer(strategy='ddp', devices>1): start one process per GPU (torchrun / "
                                   "torch.distributed.run) and initialise torch.distributed first; this Trainer does not fork")
            return
        self.world_size = dist.get_world_size(self.process_group)
        self.global_rank = dist.get_rank(self.process_group)

    @property
    def is_global_zero(self):
        return self.global_rank == 0

    def _shard(self, loader, epoch, shuffle_seed_offset=0):
        """Lightning replaces a DataLoader's sampler by DistributedSampler under DDP.  Loaders that can re-shard
        themselves (`.shard(rank, world, epoch, seed)`, e.g. scripts/pretrain_pipeline.py's HBM-table loaders) do;
        any other iterable is passed through unchanged (every rank then sees all of it — also Lightning's behaviour for
        iterables it cannot wrap)."""
        if self.world_size > 1 and hasattr(loader, "shard"):
            return loader.shard(self.global_rank, self.world_size, epoch, self.seed + shuffle_seed_offset)
        return loader

    def _reduce_mean(self, value, device):
        """mean over ranks of a host scalar (val_loss: every rank must take the same checkpoint / early-stop decision)"""
        if self.world_size == 1:
            return value
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend(self.process_group) == "nccl" else "cpu")
        dist.all_reduce(t, group=self.process_group)
        return float(t.item()) / self.world_size

    def _check_deferred(self, module, device):
        """Raise a pending out-of-range-label error — on EVERY rank when any rank saw one: a rank that raised alone would leave the
        others waiting in the next collective."""
        err = None
        try:
            module.model.check_deferred_errors()
        except IndexError as ex:
            err = ex
        if self.world_size > 1 and self._reduce_mean(1.0 if err is not None else 0.0, device) > 0.0 and err is None:
            err = IndexError("another data-parallel rank saw a source / class label out of range")
        if err is not None:
            raise err

    def _to_device(self, batch, device):
        return tuple(t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)

    def _log(self, rec):
        self.history.append(rec)
        if self.logger_path and self.is_global_zero:
            with open(self.logger_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    def validate(self, module, loader, limit=None):
        module.eval()
        losses = []
        for i, batch in enumerate(self._shard(loader, 0)):
            if limit is not None and i >= limit:
                break
            loss = module.validation_step(self._to_device(batch, self._dev(module)), i)
            losses.append(loss.value)
        module.on_validation_epoch_end()
        self._check_deferred(module, self._dev(module))
        module.train()
        val = float(torch.stack(losses).double().mean()) if losses else 0.0
        return self._reduce_mean(val, self._dev(module))

    def _dev(self, module):
        if self.device is not None:
            return self.device
        return module.model._any_engine().device

    def save_checkpoint(self, module, path):
        if self.is_global_zero:              # replicas are identical: rank 0 writes (its BatchNorm running statistics, as Lightning does)
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            torch.save({"state_dict": {k: v.cpu() for k, v in module.state_dict().items()},
                        "optimizer_states": [module.optimizer.state_dict()],
                        "epoch": self.current_epoch, "global_step": self.global_step}, path)
        if self.world_size > 1:              # the other ranks may load the file right after fit()
            import torch.distributed as dist
            dist.barrier(group=self.process_group)

    def prepare(self, module):
        """The settings `fit` applies to the module before its first step (each may re-lower the model's engines): Lightning's
        gradient_clip_val / deterministic / precision / strategy."""
        module.trainer = self
        self._setup_distributed()
        if self.deterministic and not module.model.deterministic:
            module.model.deterministic = True
            module._apply_cfg()
        module.set_gradient_clip(self.gradient_clip_val)
        module.model.set_parallel(self.world_size, self.process_group, self.sync_batchnorm)
        module.model.set_precision(self.precision)

    def fit(self, module, train_dataloaders, val_dataloaders=None):
        self.prepare(module)
        # The asynchronous mode (device-side label flag, per-step losses kept on the device) is a property of THIS loop,
        # which raises the deferred errors at every epoch end.  Outside it — `model(...)`, `get_embeddings` after fit() —
        # nobody would, so both settings are restored on the way out: labels are range-checked on the host again
        # (IndexError at once, like nn.Embedding, hippie/model.py:65-66).
        saved = (module.sync_every_step, module.model.label_check)
        module.sync_every_step = self.sync_every_step
        module.model.label_check = "sync" if self.sync_every_step else "deferred"
        try:
            return self._fit(module, train_dataloaders, val_dataloaders)
        finally:
            module.sync_every_step, module.model.label_check = saved
            module.model.set_parallel(1, None, False)      # user-level calls after fit() are single-process again
            module.model.check_deferred_errors()

    def _fit(self, module, train_dataloaders, val_dataloaders):
        dev = self._dev(module)
        if self.world_size > 1:
            module.model.broadcast_from_rank0()            # DDP: every replica starts from rank 0's parameters / buffers / optimiser state
        if val_dataloaders is not None and self.num_sanity_val_steps:
            self.validate(module, val_dataloaders, self.num_sanity_val_steps)
        bad_epochs = 0
        for epoch in range(self.max_epochs):
            self.current_epoch = module.current_epoch = epoch
            module.train()
            t0 = time.perf_counter()
            n = 0
            D = 0 if self.sync_every_step else self.run_ahead
            ring = [None] * max(D, 1)
            for i, batch in enumerate(self._shard(train_dataloaders, epoch)):
                if D and ring[i % D] is not None:
                    ring[i % D].synchronize()              # step i - D is done: the host stays at most D steps ahead
                batch = self._to_device(batch, dev)
                module.optimizer.zero_grad()
                loss = module.training_step(batch, i)
                loss.backward()
                module.optimizer.step()
                if D:
                    ring[i % D] = torch.cuda.Event()
                    ring[i % D].record(torch.cuda.current_stream(dev))
                self.global_step += 1
                n += batch[0].shape[0]
            torch.cuda.current_stream(dev).synchronize()      # this fit's stream only: a concurrent fit (fit_concurrently) keeps running
            dt = time.perf_counter() - t0
            self._check_deferred(module, dev)
            module.on_train_epoch_end()
            rec = {"epoch": epoch, "train_samples_per_s": n * self.world_size / dt, "world_size": self.world_size}
            rec.update({k: float(v) for k, v in module.logged.items() if k.startswith("train")})
            if val_dataloaders is not None:
                val = self.validate(module, val_dataloaders)
                rec[self.monitor] = val
                if val < self.best_score:
                    self.best_score, bad_epochs = val, 0
                    if self.enable_checkpointing:
                        self.best_model_path = os.path.join(self.root, f"epoch={epoch}-step={self.global_step}.ckpt")
                        self.save_checkpoint(module, self.best_model_path)
                else:
                    bad_epochs += 1
            self._log(rec)
            if val_dataloaders is not None and bad_epochs >= self.patience:
                break
        return self


class _Frozen:
    """A loader whose iterators were created — and their index draws made — ahead of time, in a prescribed order: iteration k
    of the loop replays the k-th pre-drawn pass.  Batches are still gathered lazily, on the consuming thread's stream."""

    def __init__(self, passes):
        self.passes, self.k = list(passes), 0

    def __iter__(self):
        if self.k >= len(self.passes):
            raise RuntimeError("a pre-drawn loader was iterated more often than planned")
        p = self.passes[self.k]
        self.k += 1
        return iter(p)

    def __len__(self):
        return len(self.passes[min(self.k, len(self.passes) - 1)])


def _freeze(loader):
    """One pass of `loader` with its random draws made NOW.  Loaders that can pre-draw their index batches (`.frozen()`:
    scripts/pretrain_pipeline.py's HBM-table loaders create the torch DataLoader iterator over their index list and exhaust
    it) keep gathering lazily; any other iterable is materialised."""
    return loader.frozen() if hasattr(loader, "frozen") else list(loader)


def fit_concurrently(jobs):
    """Run several independent fits at the same time, each on its own HIP stream, with the BATCHES of the sequential program.
    jobs: [(trainer, module, train_loader, val_loader), ...] in the order a sequential script would fit them.

    The reference trains the wave cVAE, then the time cVAE (scripts/train_model_with_multimodal.py:208,224): two independent
    models of ~165 small launches per step each, which together fill the GPU far better than one after the other (bench.py:
    two streams 4.4 ms per pair-step, back to back 5.9 ms).  What couples the two fits in the reference is torch's global
    generators.  The CPU one: every DataLoader iterator draws a base seed, a shuffling one also its permutation seed, in program
    order (sanity validation, then per epoch: train pass, validation pass; all of model 1 before model 2) — here every pass of
    every job is pre-drawn in exactly that order BEFORE the first step runs (_freeze), then each job's ordinary `Trainer.fit`
    runs on a thread and stream of its own over its pre-drawn passes: same batches as the sequential program.  The DEVICE one:
    the reparameterisation noise (`torch.randn_like`, hippie/model.py:48).  Two threads drawing from the one global device
    generator would consume it in a scheduling-dependent order, so every network without a prescribed noise source
    (`set_eps_source`) gets a device generator of its own for the duration of the fit, seeded here — before the threads start, in
    job order — from the global device generator: a seeded run (`torch.manual_seed`) is reproducible, and with prescribed noise
    the numbers are the sequential program's (tests/test_gpu_pipeline.py); with self-drawn noise they are NOT the sequential
    program's draws (that one stream of noise cannot be split over two concurrent consumers), only equally distributed.

    Exactness needs the number of passes to be known up front: early stopping cannot trigger when max_epochs <= patience (the
    scripts' defaults: 1 epoch, patience 30).  Otherwise, and for a single job, the fits simply run one after the other."""
    import threading
    jobs = list(jobs)
    if len(jobs) < 2 or any(val is not None and tr.max_epochs > tr.patience for tr, _, _, val in jobs):
        for tr, mod, train, val in jobs:
            tr.fit(mod, train, val)
        return [tr for tr, _, _, _ in jobs]
    planned = []
    for tr, mod, train, val in jobs:                       # the sequential program's draw order
        vals, trains = [], []
        if val is not None and tr.num_sanity_val_steps:
            vals.append(_freeze(val))
        for _ in range(tr.max_epochs):
            trains.append(_freeze(train))
            if val is not None:
                vals.append(_freeze(val))
        planned.append((_Frozen(trains), _Frozen(vals) if val is not None else None))
    # Everything the threads will replay is lowered and captured NOW, one job after the other: stream capture does not tolerate
    # another thread's launches (hippie_amd.program._CAPTURE_LOCK), and lowering is host work better not interleaved either.
    cal = []                                               # one engine per job for the stream-pair measurement
    for (tr, mod, _, _), (trains, vals) in zip(jobs, planned):
        tr.prepare(mod)
        if not mod.model.use_graph:
            continue
        for passes, segs in ((trains, ("fwd_train", "bwd", "bwd_dec", "bwd_enc", "opt")), (vals, ("fwd_eval",))):
            shapes = set()
            for batch in (passes.passes[0] if passes is not None and passes.passes else ()):
                shapes.add((int(batch[0].shape[0]), batch[-1].ndim == 2))
            for B_, with_class in shapes:
                mod.model.engine(B_, with_class).capture_segments(segs)
                if passes is trains and len(cal) < len(planned) and (not cal or cal[-1][0] is not mod):
                    cal.append((mod, mod.model.engine(B_, with_class)))
    errors = [None] * len(jobs)
    dev = jobs[0][0]._dev(jobs[0][1])
    main = torch.cuda.current_stream(dev)
    if len(cal) == len(jobs):
        # which pairs of HIP streams really overlap is a lottery of hardware-queue assignment (hippie_amd/streams.py): measured, on
        # the evaluation-forward graphs (they change no state), before the first step
        from .streams import pick_concurrent_streams
        streams = pick_concurrent_streams([e for _, e in cal], dev)
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in jobs]

    # a noise generator per network that draws its own (see the docstring): seeds from the global device generator, in job order
    own_gen = []
    with torch.cuda.device(dev):
        seeds = torch.randint(0, 2 ** 62, (len(jobs),), device=dev, dtype=torch.int64).tolist()
    for (tr, mod, _, _), seed in zip(jobs, seeds):
        net = mod.model
        if getattr(net, "eps_source", None) is None and getattr(net, "eps_generator", None) is None and hasattr(net, "set_eps_generator"):
            g = torch.Generator(device=dev)
            g.manual_seed(int(seed))
            net.set_eps_generator(g)
            own_gen.append(net)

    def run(k):
        tr, mod, _, _ = jobs[k]
        try:
            with torch.cuda.device(dev), torch.cuda.stream(streams[k]):
                streams[k].wait_stream(main)
                tr.fit(mod, planned[k][0], planned[k][1])
        except BaseException as ex:          # re-raised on the calling thread
            errors[k] = ex

    threads = [threading.Thread(target=run, args=(k,), name=f"fit-{k}") for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s_ in streams:
        main.wait_stream(s_)
    for net in own_gen:
        net.set_eps_generator(None)
    for ex in errors:
        if ex is not None:
            raise ex
    return [tr for tr, _, _, _ in jobs]
