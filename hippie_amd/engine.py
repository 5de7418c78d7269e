"""Device-side engine: owns the six arenas (as torch tensors = plain HBM allocations),
binds a lowered program to them and runs / graph-replays it.

torch is plumbing here (device memory, streams, checkpoint tensors); every
arithmetic op of the hot path runs inside libhippie_hip.so.  There is no CPU
fallback: constructing an Engine without a GPU or without the built library
raises.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import planner, program as P
from .program import DeviceProgram, HipEngineError


class Engine:
    def __init__(self, cfg: planner.ModelCfg, batch: int, train: planner.TrainCfg = None, with_class=False,
                 device=None, share_params_from: "Engine" = None):
        if not torch.cuda.is_available():
            raise HipEngineError("hippie_amd.Engine needs an MI355X (torch.cuda.is_available() is False); no CPU fallback")
        P.load_library()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.cfg, self.B, self.with_class = cfg, batch, with_class
        self.train_cfg = train or planner.TrainCfg()
        self._graphs = {}
        self._sync_cache = {}
        self.sync_group = None          # process group of the sync-BatchNorm collectives (None = WORLD)
        # Provider of the reparameterisation noise when set_inputs() is not handed one: None = torch's device generator
        # (`torch.randn_like(std)`, hippie/model.py:48); a callable(engine) -> [B, z] tensor lets a caller run a whole
        # pipeline (Trainer.fit, scripts) on a prescribed noise sequence — the pipeline parity test does.  Per engine
        # (the model containers hand their own `eps_source` down): nothing process-wide.
        self.eps_source = None
        # torch.Generator on this device for the noise drawn here when nothing is handed in (None: torch's global device generator).
        # Concurrent fits give every network a generator of its own: two threads drawing from the ONE global generator would consume
        # its Philox offset in a scheduling-dependent order (trainer.fit_concurrently).
        self.eps_generator = None
        self.plan = planner.lower(cfg, batch, self.train_cfg, with_class)
        self.ops = self.plan.ops.array()
        n = self.plan.n_param_floats
        with torch.cuda.device(self.device):
            if share_params_from is not None:
                o = share_params_from
                assert o.plan.n_param_floats == n and list(o.plan.params) == list(self.plan.params)
                self.params, self.grads, self.m, self.v, self.bufs = o.params, o.grads, o.m, o.v, o.bufs
                self.num_batches_tracked = o.num_batches_tracked
            else:
                self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
                self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
                self.m = torch.zeros(n, dtype=torch.float32, device=self.device)
                self.v = torch.zeros(n, dtype=torch.float32, device=self.device)
                self.bufs = torch.zeros(self.plan.n_buf_floats, dtype=torch.float32, device=self.device)
                self.num_batches_tracked = {k: 0 for k in self.plan.bn_keys}
            self.ws = torch.zeros(self.plan.ws_bytes, dtype=torch.uint8, device=self.device)
        arenas = [self.ws, self.params, self.grads, self.bufs, self.m, self.v]
        self.prog = DeviceProgram(self.ops, [a.data_ptr() for a in arenas], [a.numel() * a.element_size() for a in arenas])
        if share_params_from is None:
            self._init_bn_defaults()

    # ---- views -------------------------------------------------------------------
    def _init_bn_defaults(self):
        """BatchNorm weight = 1, running_var = 1 (torch defaults); everything else zero until loaded."""
        for k, info in self.plan.bufs.items():
            if k.endswith("running_var"):
                self.bufs[info.offset: info.offset + info.numel] = 1.0

    def io(self, name):
        ref, shape, dt = self.plan.io[name]
        tdt = {"f4": torch.float32, "i8": torch.int64, "f8": torch.float64}[dt]
        nb = int(np.prod(shape)) * (4 if dt == "f4" else 8)
        arena = self.ws if ref.space == P.WS else self.bufs.view(torch.uint8)
        return arena[ref.offset: ref.offset + nb].view(tdt).view(*shape)

    def ws_f32(self, ref, n):
        return self.ws[ref.offset: ref.offset + 4 * n].view(torch.float32)

    def param_view(self, key, arena=None):
        """torch-layout view ([Cout, Cin, k] for conv weights) of one parameter inside an arena."""
        info = self.plan.params[key]
        a = self.params if arena is None else arena
        flat = a[info.offset: info.offset + info.numel]
        if info.layout == "tnc":
            co, ci, k = info.shape
            return flat.view(k, co, ci).permute(1, 2, 0)
        return flat.view(*info.shape)

    # ---- checkpoint surface (reference state_dict keys, hippie/model.py modules) -------
    def state_dict(self, prefix=""):
        sd = OrderedDict()
        bn_of = {}
        for p in self.plan.bn_keys:
            bn_of[p + ".bias"] = p
        for k in self.plan.params:
            sd[prefix + k] = self.param_view(k).contiguous().clone()
            if k in bn_of:
                p = bn_of[k]
                for suffix in (".running_mean", ".running_var"):
                    info = self.plan.bufs[p + suffix]
                    sd[prefix + p + suffix] = self.bufs[info.offset: info.offset + info.numel].clone()
                sd[prefix + p + ".num_batches_tracked"] = torch.tensor(self.num_batches_tracked[p], dtype=torch.int64)
        return sd

    def load_state_dict(self, sd, strict=True, prefix=""):
        missing, unexpected = [], []
        known = set()
        for k, info in self.plan.params.items():
            known.add(prefix + k)
            if prefix + k not in sd:
                missing.append(k)
                continue
            src = sd[prefix + k]
            if tuple(src.shape) != tuple(info.shape):
                raise ValueError(f"size mismatch for {k}: checkpoint {tuple(src.shape)} vs model {tuple(info.shape)}")
            self.param_view(k).copy_(src.to(device=self.device, dtype=torch.float32))
        for k, info in self.plan.bufs.items():
            known.add(prefix + k)
            if prefix + k not in sd:
                missing.append(k)
                continue
            self.bufs[info.offset: info.offset + info.numel].copy_(sd[prefix + k].to(device=self.device, dtype=torch.float32).reshape(-1))
        for p in self.plan.bn_keys:
            k = prefix + p + ".num_batches_tracked"
            known.add(k)
            if k in sd:
                self.num_batches_tracked[p] = int(sd[k])
        unexpected = [k for k in sd if k.startswith(prefix) and k not in known]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}… unexpected {unexpected[:5]}…")
        return missing, unexpected

    def reset_optimizer_state(self):
        """A fresh torch.optim.AdamW: exp_avg = exp_avg_sq = 0, step = 0 (and the schedule-free scalars)."""
        self.m.zero_()
        self.v.zero_()
        self.io("adam_step").zero_()
        self.io("sf_state").zero_()

    def grad_dict(self):
        return OrderedDict((k, self.param_view(k, self.grads).contiguous().clone()) for k in self.plan.params)

    # ---- inputs ------------------------------------------------------------------
    def check_labels(self, src, cls=None):
        """nn.Embedding raises IndexError for an index outside its table (hippie/model.py:65-66); so does this
        engine, before anything reaches the kernels (which additionally treat such a row as zeros).  One small
        reduction + one host sync per call; callers that have validated their label tables once (Trainer.fit)
        pass validate=False to set_inputs."""
        for name, t, rows in (("source", src, self.cfg.num_sources), ("class", cls, self.cfg.num_classes)):
            if t is None or t.numel() == 0:
                continue
            lo, hi = (int(v) for v in torch.aminmax(t))
            if lo < 0 or hi >= rows:
                raise IndexError(f"{name} label out of range: values span [{lo}, {hi}] but the {name} embedding has {rows} rows")

    def flag_bad_labels(self, src, cls=None):
        """check_labels without the host sync: out-of-range labels are recorded in a device-side flag (the kernels treat
        such rows as zeros meanwhile) and raise_if_bad_labels() reports them later — once per epoch in Trainer.fit."""
        if getattr(self, "_bad_labels", None) is None:
            self._bad_labels = torch.zeros((), dtype=torch.bool, device=self.device)
        for t, rows in ((src, self.cfg.num_sources), (cls, self.cfg.num_classes)):
            if t is not None and t.numel():
                self._bad_labels |= ((t < 0) | (t >= rows)).any()

    def raise_if_bad_labels(self):
        bad = getattr(self, "_bad_labels", None)
        if bad is not None and bool(bad):
            self._bad_labels = None
            raise IndexError(f"a source / class label was out of range (embedding tables have {self.cfg.num_sources} / "
                             f"{self.cfg.num_classes} rows)")

    def set_inputs(self, x, src, cls=None, eps=None, x2=None, validate=True):
        """validate: True = check the labels now (one host sync), "deferred" = flag_bad_labels, False = not at all."""
        if validate == "deferred":
            self.flag_bad_labels(src, cls if self.with_class else None)
        elif validate:
            self.check_labels(src, cls if self.with_class else None)
        self.io("x").copy_(x.reshape(self.io("x").shape), non_blocking=True)
        if x2 is not None:
            self.io("x2").copy_(x2.reshape(self.io("x2").shape), non_blocking=True)
        self.io("src").copy_(src, non_blocking=True)
        if cls is not None:
            if not self.with_class:
                raise ValueError("engine was lowered without class labels")
            self.io("cls").copy_(cls, non_blocking=True)
        elif self.with_class:
            raise ValueError("engine was lowered with class labels; pass cls")
        if eps is None and self.eps_source is not None:
            eps = self.eps_source(self)
        if eps is None:
            self.io("eps").normal_(generator=self.eps_generator)          # torch.randn_like(std), hippie/model.py:48
        else:
            self.io("eps").copy_(eps, non_blocking=True)

    # ---- execution ----------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- sync-BatchNorm (TrainCfg.sync_bn_world > 1) ---------------------------------------
    def _sync_points(self, first, count):
        key = (first, count)
        pts = self._sync_cache.get(key)
        if pts is None:
            pts = []
            for k in range(first, first + count):
                if int(self.ops[k]["op"]) == P.STATS_SYNC:
                    ref = int(self.ops[k]["buf"][0])
                    assert (ref >> 56) == P.WS
                    off, n = ref & ((1 << 56) - 1), int(self.ops[k]["i"][0])
                    pts.append((k, self.ws[off: off + 8 * n].view(torch.float64)))
            self._sync_cache[key] = pts
        return pts

    def _run_synced(self, first, count):
        """Run an op range, summing every HP_OP_STATS_SYNC slot over the data-parallel ranks where its marker
        stands (torch.nn.SyncBatchNorm semantics: all ranks normalise with the statistics of the global batch).
        Eager and collective-bound (~85 small all-reduces per pass): the parity mode, not the fast one."""
        import torch.distributed as dist
        W = self.train_cfg.sync_bn_world
        if not dist.is_initialized() or dist.get_world_size(self.sync_group) != W:
            raise HipEngineError(f"sync-BatchNorm was lowered for {W} ranks; initialise torch.distributed with that world size")
        cur = first
        for k, slot in self._sync_points(first, count):
            if k > cur:
                self.prog.run(cur, k - cur, self._stream())
            dist.all_reduce(slot, op=dist.ReduceOp.SUM, group=self.sync_group)
            cur = k + 1
        if first + count > cur:
            self.prog.run(cur, first + count - cur, self._stream())

    def run(self, seg, use_graph=False):
        first, count = self.plan.ops.segments[seg]
        if count == 0:
            return
        if self.train_cfg.sync_bn_world > 1 and self._sync_points(first, count):
            return self._run_synced(first, count)
        if use_graph:
            g = self._graphs.get(seg)
            if g is None:
                g = self._graphs[seg] = self.prog.capture(first, count)
            self.prog.replay(g, self._stream())
        else:
            self.prog.run(first, count, self._stream())

    def capture_segments(self, names=("fwd_train", "bwd", "opt", "fwd_eval", "enc_eval")):
        """Capture the named segments into hipGraphs now (they are otherwise captured on first use)."""
        for seg in names:
            if seg in self.plan.ops.segments and seg not in self._graphs and self.plan.ops.segments[seg][1] > 0:
                first, count = self.plan.ops.segments[seg]
                if self.train_cfg.sync_bn_world > 1 and self._sync_points(first, count):
                    continue                     # runs eagerly around its collectives
                self._graphs[seg] = self.prog.capture(first, count)

    def forward(self, training=True, use_graph=False):
        mode = "train" if training else "eval"
        self.run("fwd_" + mode, use_graph)
        if training:
            for p in self.num_batches_tracked:
                self.num_batches_tracked[p] += 1
        return self._outputs(mode)

    def _outputs(self, mode):
        z = self.cfg.z_dim
        mulv = self.io("mulv_" + mode)
        outs = [self.io("enc_" + mode), mulv[:, :z], mulv[:, z:], self.io("rec_" + mode)]
        if self.cfg.kind == "multimodal":
            outs.append(self.io("rec2_" + mode))
        return tuple(outs)

    def encode(self, use_graph=False):
        """Eval-mode encoder half only ("enc_eval", a prefix of "fwd_eval"): (enc, mu, logvar) of
        hippieUnimodalCVAE.encode (hippie/model.py:51-57) without running the decoder."""
        self.run("enc_eval", use_graph)
        z = self.cfg.z_dim
        mulv = self.io("mulv_eval")
        return self.io("enc_eval"), mulv[:, :z], mulv[:, z:]

    def backward(self, use_graph=False):
        """Backward pass.  Exactly one per training forward: its BatchNorm reductions accumulate into the fp64
        statistic slots that the forward's first op zeroes, so a second backward would double them."""
        self.run("bwd", use_graph)

    def optimizer_step(self, use_graph=False):
        self.run("opt", use_graph)

    def optimizer_swap(self, to_eval: bool, use_graph=False):
        """AdamWScheduleFree.eval() / .train() (hippie/optimizers.py:82-103): move the parameter arena between
        the training point y and the averaged point x.  The caller tracks which mode it is in."""
        if self.train_cfg.optimizer != "schedulefree":
            raise HipEngineError("optimizer_swap needs TrainCfg(optimizer='schedulefree')")
        self.run("sf_eval" if to_eval else "sf_train", use_graph)

    def train_step(self, use_graph=False):
        """forward(train) + backward + AdamW on the staged inputs; returns the scalars tensor (device).  With graphs (and
        no collectives inside the step) the three segments replay as ONE graph."""
        if use_graph and "step" in self.plan.ops.segments and self.train_cfg.sync_bn_world <= 1:
            self.run("step", True)
            for p in self.num_batches_tracked:
                self.num_batches_tracked[p] += 1
            return self.io("scalars")
        self.forward(True, use_graph)
        self.backward(use_graph)
        self.optimizer_step(use_graph)
        return self.io("scalars")

    # ---- HBM-resident training tables (TrainCfg.resident_units) --------------------------------------------------------
    def load_dataset(self, x, labels, x2=None, perm=None, seed=0):
        """Copy the preprocessed training tables ([N, L] per modality, int64 source labels [N]) into the engine's workspace, set
        the permutation the staged steps walk through (default: identity) and the noise seed; the batch cursor starts at 0."""
        N = self.train_cfg.resident_units
        if N <= 0:
            raise HipEngineError("load_dataset needs TrainCfg(resident_units=N)")
        self.io("data_x").copy_(x.reshape(N, -1))
        if x2 is not None:
            self.io("data_x2").copy_(x2.reshape(N, -1))
        lo, hi = (int(v) for v in torch.aminmax(labels))
        if lo < 0 or hi >= self.cfg.num_sources:
            raise IndexError(f"source label out of range: values span [{lo}, {hi}] but the source embedding has {self.cfg.num_sources} rows")
        self.io("data_labels").copy_(labels)
        self.set_permutation(torch.arange(N) if perm is None else perm)
        self.io("seed").fill_(int(seed))
        self.io("cursor").zero_()

    def set_permutation(self, perm):
        """The order in which staged steps visit the units (a new shuffle per epoch is a new call, or a longer cursor walk over
        one permutation: batch j of the walk is perm[j*B:(j+1)*B], wrapping after N // (B * dp_world) batches per rank)."""
        p = self.io("perm")
        if perm.numel() != p.numel():
            raise ValueError(f"the permutation must list all {p.numel()} resident units")
        perm = perm.to(torch.int64)
        lo, hi = (int(v) for v in torch.aminmax(perm))
        if lo < 0 or hi >= p.numel():          # (the kernel would read row 0 for such an entry: never a fault, but never what was meant)
            raise IndexError(f"permutation entries span [{lo}, {hi}] but there are {p.numel()} resident units")
        p.copy_(perm)

    def train_step_staged(self, use_graph=True):
        """One optimisation step on the next batch of the resident tables: HP_OP_STAGE_BATCH (index gather + Philox eps) +
        forward + backward + optimiser as ONE graph replay where nothing sits between them, else stage+forward | backward | opt
        (the caller all-reduces between `backward` and `optimizer_step` itself: see staged_forward)."""
        if "stage" not in self.plan.ops.segments:
            raise HipEngineError("train_step_staged needs TrainCfg(resident_units=N)")
        if self.train_cfg.dp_world > 1:          # a replica that skipped the gradient all-reduce would silently diverge from the others
            raise HipEngineError(f"this engine was lowered for {self.train_cfg.dp_world} data-parallel ranks: use staged_forward(), "
                                 "parallel.backward_allreduce(engine, group), optimizer_step()")
        if use_graph and "step_staged" in self.plan.ops.segments and self.train_cfg.sync_bn_world <= 1:
            self.run("step_staged", True)
        else:
            self.run("stage", use_graph)
            self.run("fwd_train", use_graph)
            self.run("bwd", use_graph)
            self.run("opt", use_graph)
        for p in self.num_batches_tracked:
            self.num_batches_tracked[p] += 1
        return self.io("scalars")

    def staged_forward(self, use_graph=True):
        """stage + training forward only (data-parallel callers: backward -> gradient all-reduce -> optimizer_step follow)"""
        if use_graph and "fwd_train_staged" in self.plan.ops.segments and self.train_cfg.sync_bn_world <= 1:
            self.run("fwd_train_staged", True)
            for p in self.num_batches_tracked:
                self.num_batches_tracked[p] += 1
            return self._outputs("train")
        self.run("stage", use_graph)
        return self.forward(True, use_graph)

    def scalars(self):
        """(loss, mse1, mse2, kl_mean) of the last forward, synchronising (= the reference's loss.item())."""
        return [float(v) for v in self.io("scalars").tolist()]

    @property
    def adam_step(self):
        return int(self.io("adam_step")[0])

    def profile(self, seg):
        first, count = self.plan.ops.segments[seg]
        ms = self.prog.profile(first, count, self._stream())
        return [(P.OP_NAMES[int(self.ops[first + k]["op"])], self.plan.ops.notes[first + k], float(ms[k])) for k in range(count)]
