"""The reference's class surface (hippie/model.py) on top of the MI355X engine.

Same constructor signatures, forward return tuples, logged metric names, `.model`,
`.optimizer`, and `state_dict()` keys as the reference, so the scripts' call pattern
(scripts/train_model_with_multimodal.py:169-230, scripts/utils.py:75-101) carries over:

    net = hippieUnimodalCVAE(z_dim=10, output_size=50, class_hidden_dim=5, num_sources=5, num_classes=5)
    module = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3, weight_decay=0.01)
    Trainer(max_epochs=..., gradient_clip_val=...).fit(module, train_loader, val_loader)
    enc, mu, logvar, dec = module((x, labels))

Engines are lowered per (batch size, label mode) on first use and share one set of parameter /
optimiser arenas.  All arithmetic runs in libhippie_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import replace

import torch

from . import planner
from .engine import Engine


class _Net:
    """Common machinery of the two cVAE containers (not an nn.Module: parameters live in HBM arenas)."""

    kind = "unimodal"

    def _init(self, cfg: planner.ModelCfg, device=None):
        self.cfg = cfg
        self.device = device
        self.training = True
        self._engines = {}            # (B, with_class) -> Engine
        self._root = None
        self._train_cfg = planner.TrainCfg()
        self.deterministic = False           # see configure_training
        # Parameters are drawn HERE, at construction, from torch's global CPU generator in the reference constructor's
        # order: `torch.manual_seed(42); ...; wave = hippieUnimodalCVAE(...); time = hippieUnimodalCVAE(...)` consumes
        # the generator exactly as the reference script does (scripts/train_model_with_multimodal.py:78,169-176).  They
        # move to HBM when the first engine is lowered.
        self._pending_sd = reference_init_state(cfg)
        self._generation = 0
        # hipGraph replay of the lowered segments (one capture per batch shape / label mode): what Trainer.fit and the
        # scripts run.  False = eager launches (debugging).
        self.use_graph = True
        # "sync": labels are range-checked on the host before every forward (IndexError at once, like nn.Embedding);
        # "deferred": device-side flag, raised by check_deferred_errors() — Trainer.fit does that once per epoch
        self.label_check = "sync"
        # callable(engine) -> [B, z] reparameterisation noise for every forward of THIS network that is not handed one
        # (None: torch's device generator); handed down to the engines
        self.eps_source = None
        self.eps_generator = None          # torch.Generator(device) for that noise (None: the global device generator); see Engine
        # data parallel (set by Trainer.fit for the duration of a fit): replicas per process group, gradient mean between
        # backward and optimizer.step (hippie_amd.parallel), optional sync-BatchNorm; and the MFMA operand precision
        self.dp_world, self.dp_group, self.sync_batchnorm = 1, None, False
        # True: under DDP the backward pass runs as two halves with the decoder-side gradient bucket all-reduced on a side stream under the
        # encoder-side half (TrainCfg.bucketed_bwd).  Off by default: on this runtime the cross-stream event dependencies between the
        # graph launches cost more than the transfer they hide (1-rank ratio 0.78 / 0.95 / 0.97 against 0.98 / 0.98 / 1.00 for one
        # collective after the pass — wave + time pair / multimodal at batch 512 / at batch 8192)
        self.ddp_bucketed = False
        self.precision = "f32"
        # precision "f32": which matrix cores carry the fp32 arithmetic — "bf16x3" (three-term operand split on the bf16 cores, the default of
        # planner.TrainCfg.mfma_dtype) or "f32" (v_mfma_f32_32x32x2_f32); same results to the fp32 rounding level either way
        self.fp32_matrix_path = planner.TrainCfg().mfma_dtype if planner.TrainCfg().mfma_dtype != "bf16" else "bf16x3"
        self.bf16_storage = False          # precision "bf16": True also STORES the backbones' activations as bf16 (TrainCfg.act_dtype: a third less
                                           # workspace, slower as the kernels stand); False rounds only the matrix operands

    # -- engine cache -----------------------------------------------------------------
    def engine(self, batch, with_class) -> Engine:
        key = (int(batch), bool(with_class))
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(self.cfg, key[0], self._train_cfg, with_class=key[1], device=self.device, share_params_from=self._root)
            eng.eps_source = self.eps_source
            eng.eps_generator = self.eps_generator
            eng.sync_group = self.dp_group
            if self._root is None:
                self._root = eng
                if self._pending_sd is not None:
                    eng.load_state_dict(self._pending_sd, strict=False)
                    self._pending_sd = None
                else:
                    _default_init(eng)
            self._engines[key] = eng
        return eng

    def configure_training(self, train_cfg: planner.TrainCfg, reset_optimizer=False):
        """(Re)lower the optimiser / loss constants (lr, weight decay, beta, clip, modality weights).
        Parameters and BatchNorm buffers are kept.  The optimiser state (exp_avg, exp_avg_sq, step; schedule-free
        scalars) is kept by default — re-lowerings inside one training module, e.g. set_gradient_clip — and
        ZEROED with reset_optimizer=True: a new train module over an existing network owns a fresh
        torch.optim.AdamW in the reference (hippie/model.py:93; the label-free fine-tune stage re-wraps the
        pretrained network, scripts/train_model_with_multimodal.py:263-268)."""
        # Lightning's Trainer(deterministic=True) / torch.use_deterministic_algorithms(True): the weight gradients are summed
        # in a fixed order (per-split slabs + ordered reduce) instead of with fp32 atomics — runs become bit-reproducible
        if (self.deterministic or torch.are_deterministic_algorithms_enabled()) and not train_cfg.deterministic_wgrad:
            train_cfg = replace(train_cfg, deterministic_wgrad=True)
        # network-level settings survive every re-lowering of the optimiser constants
        train_cfg = replace(train_cfg, mfma_dtype=self.fp32_matrix_path if self.precision == "f32" else "bf16", act_dtype="bf16" if (self.precision == "bf16" and self.bf16_storage) else "f32", bucketed_bwd=bool(self.ddp_bucketed) and self.dp_world > 1,
                            sync_bn_world=self.dp_world if (self.sync_batchnorm and self.dp_world > 1) else 0)
        self._train_cfg = train_cfg
        keep = self._root
        self.check_deferred_errors()          # a pending bad-label flag lives on the engines dropped below: report it first
        self._engines = {}
        if keep is not None:
            eng = Engine(self.cfg, keep.B, train_cfg, with_class=keep.with_class, device=self.device, share_params_from=keep)
            eng.eps_source = self.eps_source
            eng.eps_generator = self.eps_generator
            eng.sync_group = self.dp_group
            if reset_optimizer:
                eng.reset_optimizer_state()
            self._root = eng
            self._engines[(keep.B, keep.with_class)] = eng
        self._generation += 1

    def set_fp32_matrix_path(self, path):
        """"bf16x3" | "f32": see `fp32_matrix_path` (takes effect for precision "f32"; re-lowers, parameters and optimiser state kept)."""
        if path not in ("bf16x3", "f32"):
            raise ValueError(f"fp32_matrix_path must be 'bf16x3' or 'f32', not {path!r}")
        if path != self.fp32_matrix_path:
            self.fp32_matrix_path = path
            self.configure_training(self._train_cfg)

    def set_precision(self, precision):
        """"f32": the reference's arithmetic (the parity path; on which matrix cores: `fp32_matrix_path`).  "bf16": BASELINE config 2's reduced-precision mode —
        conv / weight-gradient operands in bfloat16 on v_mfma_f32_32x32x16_bf16, fp32 accumulation, statistics, master weights
        and AdamW (planner.TrainCfg.mfma_dtype; tolerance: tests/test_gpu_bf16.py).  Trainer(precision="bf16") calls this."""
        if precision not in ("f32", "bf16"):
            raise ValueError(f"precision must be 'f32' or 'bf16', not {precision!r}")
        if precision != self.precision:
            self.precision = precision
            self.configure_training(self._train_cfg)

    def set_parallel(self, world, group=None, sync_batchnorm=False):
        """Data-parallel replicas of this network (one process per GPU, Lightning-DDP semantics): `_Loss.backward()` then
        mean-all-reduces the gradient arena over `group` before optimizer.step(); sync_batchnorm=True re-lowers with
        sync-BatchNorm markers (torch.nn.SyncBatchNorm semantics, Lightning's sync_batchnorm=True)."""
        world = int(world)
        changed = (bool(sync_batchnorm) and world > 1) != (self._train_cfg.sync_bn_world > 1) or \
                  (bool(sync_batchnorm) and world > 1 and self._train_cfg.sync_bn_world != world) or \
                  (bool(self.ddp_bucketed) and world > 1) != bool(self._train_cfg.bucketed_bwd)
        self.dp_world, self.dp_group, self.sync_batchnorm = world, group, bool(sync_batchnorm)
        if changed:
            self.configure_training(self._train_cfg)
        for eng in self._engines.values():
            eng.sync_group = group

    def broadcast_from_rank0(self):
        """DDP start-up: parameters, BatchNorm buffers and optimiser state of every replica := rank 0's."""
        from . import parallel
        eng = self._any_engine()
        parallel.broadcast_([eng.params, eng.bufs, eng.m, eng.v], 0, self.dp_group)
        nbt = torch.tensor([eng.num_batches_tracked[k] for k in eng.plan.bn_keys], dtype=torch.int64, device=eng.device)
        parallel.broadcast_([nbt], 0, self.dp_group)
        for k, v in zip(eng.plan.bn_keys, nbt.tolist()):
            eng.num_batches_tracked[k] = int(v)

    def set_eps_source(self, fn):
        """Prescribe the reparameterisation noise of this network's forwards (None = torch's device generator)."""
        self.eps_source = fn
        for eng in self._engines.values():
            eng.eps_source = fn

    def set_eps_generator(self, gen):
        """A torch.Generator (on this network's device) for the noise its forwards draw themselves; None = the global device generator."""
        self.eps_generator = gen
        for eng in self._engines.values():
            eng.eps_generator = gen

    def _any_engine(self) -> Engine:
        if self._root is None:
            self.engine(2, False)
        return self._root

    # -- nn.Module-like surface ----------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        if self._root is not None and torch.device(device) != self._root.device:
            raise RuntimeError("parameters already live on " + str(self._root.device))
        self.device = device
        return self

    def state_dict(self, prefix=""):
        return self._any_engine().state_dict(prefix)

    def load_state_dict(self, sd, strict=True, prefix=""):
        return self._any_engine().load_state_dict(sd, strict=strict, prefix=prefix)

    # -- the embedding tables as attributes (scripts/train_model_with_multimodal.py:378-379 REPLACES one) -----------------
    class _Embedding:
        """What `model.class_embedding` / `model.source_embedding` reads as: weight (a view into the parameter arena),
        num_embeddings, embedding_dim — the attributes of the reference's nn.Embedding that its scripts touch."""

        def __init__(self, net, key):
            self._net, self._key = net, key

        @property
        def weight(self):
            eng = self._net._any_engine()
            return eng.param_view(self._key + ".weight", eng.params)

        @property
        def num_embeddings(self):
            return int(self.weight.shape[0])

        @property
        def embedding_dim(self):
            return int(self.weight.shape[1])

    @property
    def source_embedding(self):
        return _Net._Embedding(self, "source_embedding")

    @property
    def class_embedding(self):
        return _Net._Embedding(self, "class_embedding")

    @class_embedding.setter
    def class_embedding(self, module):
        """`model.class_embedding = nn.Embedding(n, class_hidden_dim)` (the reference's supervised stage, scripts/...:378-379): a
        NEW class table of n rows with the given weights; every other parameter, the BatchNorm buffers and — for the tensors whose
        shape is unchanged — the AdamW moments are kept.  The network is re-lowered for the new table size on its next use."""
        w = module.weight if hasattr(module, "weight") else module
        w = torch.as_tensor(w).detach().float()
        H = self.cfg.class_hidden_dim
        if w.ndim != 2 or w.shape[1] != H:
            raise ValueError(f"class_embedding must be [num_classes, {H}], got {tuple(w.shape)}")
        self.check_deferred_errors()
        if self._root is not None:
            eng = self._root
            sd = {k: v.detach().clone() for k, v in eng.state_dict().items()}
            moments = {k: (eng.param_view(k, eng.m).detach().clone(), eng.param_view(k, eng.v).detach().clone()) for k in eng.plan.params}
            step = eng.adam_step
        else:
            sd, moments, step = dict(self._pending_sd), None, 0
        sd["class_embedding.weight"] = w.clone()
        self.cfg = replace(self.cfg, num_classes=int(w.shape[0]))
        self._engines, self._root = {}, None
        self._pending_sd = sd
        self._generation += 1
        if moments is not None:
            eng = self._any_engine()                     # lowered for the new table; parameters from `sd`
            for k, (m_, v_) in moments.items():
                if k in eng.plan.params and tuple(m_.shape) == tuple(eng.plan.params[k].shape):
                    eng.param_view(k, eng.m).copy_(m_)
                    eng.param_view(k, eng.v).copy_(v_)
            eng.io("adam_step").fill_(step)

    def parameters(self):
        """Handle for optimisers (hippie_amd.optimizers.AdamWScheduleFree(model.parameters(), ...))."""
        from .optimizers import ParamHandle
        return ParamHandle(self)

    def parameters_numel(self):
        return sum(i.numel for i in self._any_engine().plan.params.values())

    def _run_forward(self, eng: Engine, x, src, cls, eps, x2=None):
        if self.training and eng.B < 2:
            # torch's BatchNorm1d in training mode: "Expected more than 1 value per channel when training"
            raise ValueError("Expected more than 1 value per channel when training (batch of 1 in train mode)")
        eng.set_inputs(x, src, cls, eps, x2=x2, validate=True if self.label_check == "sync" else "deferred")
        return eng.forward(training=self.training, use_graph=self.use_graph)

    def check_deferred_errors(self):
        for eng in self._engines.values():
            eng.raise_if_bad_labels()


def reference_param_order(cfg: planner.ModelCfg):
    """Parameter keys with shapes in the order the reference's constructors CREATE (and therefore randomly initialise)
    them: hippieUnimodalCVAE.__init__ (hippie/model.py:13-44), MultiModalCVAE.__init__ (:352-395), ResNet18Enc /
    BasicBlockEnc (hippie/backbones.py:20-34,74-92), ResNet18Dec / BasicBlockDec / ResizeConv1d (:7-11,45-63,107-126).
    It is also the reference's state_dict order.  kind: "conv" | "linear" (weight then bias, kaiming-uniform(a=sqrt 5)
    and U(+-1/sqrt(fan_in))), "bn" (ones / zeros, no draw), "emb" (N(0,1))."""
    z, H = cfg.z_dim, cfg.class_hidden_dim
    out = []

    def conv(key, co, ci, k, bias=False):
        out.append((key + ".weight", (co, ci, k), "weight"))
        if bias:
            out.append((key + ".bias", (co,), "bias:" + key + ".weight"))

    def linear(key, n, k):
        out.append((key + ".weight", (n, k), "weight"))
        out.append((key + ".bias", (n,), "bias:" + key + ".weight"))

    def bn(key, c):
        out.append((key + ".weight", (c,), "ones"))
        out.append((key + ".bias", (c,), "zeros"))

    def encoder(pre):
        conv(pre + "conv1", 64, 1, 3)
        bn(pre + "bn1", 64)
        cin = 64
        for li, planes in enumerate((64, 128, 256, 512), start=1):
            for bi in (0, 1):
                stride = 2 if (bi == 0 and li > 1) else 1
                p = f"{pre}layer{li}.{bi}."
                conv(p + "conv1", planes, cin, 3)
                bn(p + "bn1", planes)
                conv(p + "conv2", planes, planes, 3)
                bn(p + "bn2", planes)
                if stride != 1:
                    conv(p + "shortcut.0", planes, cin, 1)
                    bn(p + "shortcut.1", planes)
                cin = planes
        linear(pre + "linear", 2 * z, 512)

    def decoder(pre, output_size):
        linear(pre + "linear", 512, 2 * z)
        cin = 512
        for li, planes in ((4, 256), (3, 128), (2, 64), (1, 64)):
            for bi, stride in enumerate((1, 1 if li == 1 else 2)):
                p = f"{pre}layer{li}.{bi}."
                cout = cin // stride
                conv(p + "conv2", cin, cin, 3)
                bn(p + "bn2", cin)
                if stride == 1:
                    conv(p + "conv1", cout, cin, 3)
                    bn(p + "bn1", cout)
                else:
                    conv(p + "conv1.conv", cout, cin, 3, bias=True)
                    bn(p + "bn1", cout)
                    conv(p + "shortcut.0.conv", cout, cin, 3, bias=True)
                    bn(p + "shortcut.1", cout)
            cin = planes
        conv(pre + "conv1.conv", 1, 64, 3, bias=True)
        linear(pre + "linear_out", output_size, 64)

    def dec_fc(name):
        linear(name + ".0", 2 * z, z + 2 * H)
        linear(name + ".2", 2 * z, 2 * z)
        bn(name + ".3", 2 * z)

    if cfg.kind == "unimodal":
        encoder("encoder.")
        linear("encoder_fc.0", 2 * z, 2 * z + 2 * H)
        bn("encoder_fc.1", 2 * z)
        linear("encoder_fc.3", z, 2 * z)
        bn("encoder_fc.4", z)
    else:
        encoder("encoder_mod1.")
        encoder("encoder_mod2.")
        linear("fusion_encoder.0", 2 * z, 4 * z + 2 * H)
        bn("fusion_encoder.1", 2 * z)
        linear("fusion_encoder.3", z, 2 * z)
    out.append(("source_embedding.weight", (cfg.num_sources, H), "emb"))
    out.append(("class_embedding.weight", (cfg.num_classes, H), "emb"))
    linear("z_mean", z, z)
    linear("z_log_var", z, z)
    if cfg.kind == "unimodal":
        dec_fc("decoder_fc")
        decoder("decoder.", cfg.output_size)
    else:
        dec_fc("decoder_fc_mod1")
        dec_fc("decoder_fc_mod2")
        decoder("decoder_mod1.", cfg.output_size)
        decoder("decoder_mod2.", cfg.output_size2)
    return out


def reference_init_state(cfg: planner.ModelCfg, generator=None):
    """What the reference's constructor leaves in its parameters, drawn from torch's CPU generator (the global one by
    default) with the same torch.nn.init calls in the same order — `torch.manual_seed(42)` followed by the same
    preceding draws therefore gives the reference's initial weights bit for bit (pinned by tests/golden/init_seed42.npz,
    produced by the reference classes).  nn.Conv1d / nn.Linear.reset_parameters: kaiming_uniform_(weight, a=sqrt(5)),
    bias ~ U(+-1/sqrt(fan_in)); nn.Embedding: N(0, 1); BatchNorm1d: weight 1, bias 0 (no draw)."""
    import math
    sd = OrderedDict()
    for key, shape, kind in reference_param_order(cfg):
        t = torch.empty(shape, dtype=torch.float32)
        if kind == "weight":
            torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5), generator=generator)
        elif kind.startswith("bias:"):
            w = sd[kind[5:]]
            fan_in = w[0].numel()                       # in_channels * kernel_size (Conv1d) / in_features (Linear)
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            torch.nn.init.uniform_(t, -bound, bound, generator=generator)
        elif kind == "emb":
            torch.nn.init.normal_(t, generator=generator)
        elif kind == "ones":
            t.fill_(1.0)
        else:
            t.zero_()
        sd[key] = t
    return sd


def _default_init(eng: Engine, seed=None):
    """The reference constructor's initialisation (reference_init_state), from the global CPU generator unless a seed
    is given."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    sd = reference_init_state(eng.cfg, g)
    assert set(sd) == set(eng.plan.params), set(sd) ^ set(eng.plan.params)
    eng.load_state_dict(sd, strict=False)


class hippieUnimodalCVAE(_Net):
    """hippie/model.py:12-72."""

    def __init__(self, z_dim, output_size, class_hidden_dim, num_sources, num_classes, device=None):
        self.z_dim, self.class_hidden_dim = z_dim, class_hidden_dim
        self.num_sources, self.num_classes = num_sources, num_classes
        self._init(planner.ModelCfg("unimodal", z_dim, output_size, 0, class_hidden_dim, num_sources, num_classes), device)

    def forward(self, data, source_labels, class_labels=None, eps=None):
        """-> (encoded, mu, logvar, decoded[B,1,L]); eps (reparameterisation noise) is drawn on the device
        when not given — also in eval mode, as the reference does (model.py:46-49)."""
        if data.shape[-1] != self.cfg.output_size:
            raise ValueError(f"expected input length {self.cfg.output_size}, got {tuple(data.shape)}")
        eng = self.engine(data.shape[0], class_labels is not None)
        return self._run_forward(eng, data, source_labels, class_labels, eps)

    __call__ = forward

    def encode_labels(self, data, source_labels, class_labels=None):
        """Eval-mode encoder half only -> (encoded, mu, logvar): what `encode` (model.py:51-57) returns, with the
        embedding look-ups of `forward` (:65-66) included and the decoder skipped.  Same kernels and therefore the
        same numbers as forward()[:3]; used by the embedding path, which keeps only `encoded`."""
        if self.training:
            raise RuntimeError("encode_labels() is the eval-mode fast path; call .eval() first (train mode needs the full forward)")
        if data.shape[-1] != self.cfg.output_size:
            raise ValueError(f"expected input length {self.cfg.output_size}, got {tuple(data.shape)}")
        eng = self.engine(data.shape[0], class_labels is not None)
        eng.set_inputs(data, source_labels, class_labels, eps=torch.zeros_like(eng.io("eps")),
                       validate=True if self.label_check == "sync" else "deferred")
        return eng.encode(use_graph=self.use_graph)


class MultiModalCVAE(_Net):
    """hippie/model.py:350-432."""

    kind = "multimodal"

    def __init__(self, z_dim, output_size_wave, output_size_isi, class_hidden_dim, num_sources, num_classes, device=None):
        self.z_dim, self.class_hidden_dim = z_dim, class_hidden_dim
        self.num_sources, self.num_classes = num_sources, num_classes
        self._init(planner.ModelCfg("multimodal", z_dim, output_size_wave, output_size_isi, class_hidden_dim, num_sources, num_classes), device)

    def forward(self, data1, data2, source_labels, class_labels=None, eps=None):
        eng = self.engine(data1.shape[0], class_labels is not None)
        return self._run_forward(eng, data1, source_labels, class_labels, eps, x2=data2)

    __call__ = forward


# ======================================================================================
class _Loss:
    """What training_step returns: a scalar handle with .item() / .backward() / float()."""

    def __init__(self, eng: Engine, slot=0, use_graph=True, dp_world=1, dp_group=None):
        self.eng, self.slot, self.use_graph = eng, slot, use_graph
        self.dp_world, self.dp_group = dp_world, dp_group
        self._backward_done = False
        self.value = eng.io("scalars")[slot].clone()      # device scalar of THIS step (the slot is overwritten by the next one)

    def item(self):
        return float(self.value)                          # host sync, like torch's loss.item()

    __float__ = item

    def detach(self):
        return self.value

    def backward(self):
        if not self._backward_done:
            if self.dp_world > 1:
                # DDP: gradient MEAN over the replicas between backward and optimizer.step (Lightning's default strategy on a
                # multi-GPU host, scripts/train_model_with_multimodal.py:200-207); RCCL over xGMI when the backend is "nccl".  (With
                # net.ddp_bucketed the plan holds two backward halves and the decoder-side bucket is reduced under the encoder-side one.)
                from . import parallel
                parallel.backward_allreduce(self.eng, self.dp_group, self.use_graph)
            else:
                self.eng.backward(self.use_graph)
            self._backward_done = True


class _Optimizer:
    """The `.optimizer` attribute (model.py:93): AdamW state lives in the engine arenas."""

    def __init__(self, module):
        self.module = module
        self.last_engine = None

    def zero_grad(self, set_to_none=True):
        pass                                   # the backward program zeroes the gradient arena itself

    def step(self):
        eng = self.last_engine
        if eng is None:
            raise RuntimeError("optimizer.step() before any training_step")
        eng.optimizer_step(self.module.model.use_graph)

    @property
    def param_groups(self):
        t = self.module.model._train_cfg
        return [dict(lr=t.lr, weight_decay=t.weight_decay, betas=(t.beta1, t.beta2), eps=t.adam_eps)]

    def state_dict(self):
        eng = self.module.model._any_engine()
        state = OrderedDict()
        for i, k in enumerate(eng.plan.params):
            state[i] = dict(step=torch.tensor(float(eng.adam_step)), exp_avg=eng.param_view(k, eng.m).contiguous().clone(),
                            exp_avg_sq=eng.param_view(k, eng.v).contiguous().clone())
        return dict(state=state, param_groups=[dict(self.param_groups[0], params=list(range(len(state))))],
                    param_names=list(eng.plan.params))

    def load_state_dict(self, sd):
        eng = self.module.model._any_engine()
        names = sd.get("param_names", list(eng.plan.params))
        step = 0
        for i, k in enumerate(names):
            st = sd["state"].get(i)
            if st is None or k not in eng.plan.params or tuple(st["exp_avg"].shape) != tuple(eng.plan.params[k].shape):
                continue                      # e.g. a class_embedding re-created with another num_classes
            eng.param_view(k, eng.m).copy_(st["exp_avg"].to(eng.device))
            eng.param_view(k, eng.v).copy_(st["exp_avg_sq"].to(eng.device))
            step = max(step, int(st["step"]))
        eng.io("adam_step").fill_(step)


class _TrainModule:
    def _setup(self, base_model, learning_rate, weight_decay, beta, alpha_max, w1=1.0, w2=1.0):
        self.model = base_model
        self.alpha_max, self.beta = alpha_max, beta
        self.lr, self.weight_decay = learning_rate, weight_decay
        self.mod1_weight, self.mod2_weight = w1, w2
        self.val_loss, self.train_loss = [], []
        # The reference appends loss.item() in every step (hippie/model.py:114): one host sync per step.  With
        # sync_every_step False the per-step values stay on the device and are fetched once, when the epoch-end hooks
        # average them (same printed numbers, no per-step sync) — Trainer.fit selects that.
        self.sync_every_step = True
        self.logged = {}
        self.current_epoch = 0
        self.trainer = None
        self.gradient_clip_val = 0.0
        self.training = True
        self._apply_cfg(reset_optimizer=True)
        self.optimizer = _Optimizer(self)

    def _apply_cfg(self, reset_optimizer=False):
        opt = getattr(self, "optimizer", None)
        if opt is not None and not isinstance(opt, _Optimizer):
            # a user-installed optimiser (AdamWScheduleFree) owns lr / weight decay; keep its settings
            self.model.configure_training(replace(self.model._train_cfg, beta=self.beta, clip=self.gradient_clip_val or 0.0,
                                                  w1=self.mod1_weight, w2=self.mod2_weight))
            return
        self.model.configure_training(planner.TrainCfg(lr=self.lr, weight_decay=self.weight_decay, beta=self.beta,
                                                       clip=self.gradient_clip_val or 0.0, w1=self.mod1_weight, w2=self.mod2_weight),
                                      reset_optimizer=reset_optimizer)

    def set_gradient_clip(self, val):
        """Lightning's Trainer(gradient_clip_val=...): folded into the fused AdamW launch."""
        val = float(val or 0.0)
        if val != self.gradient_clip_val:
            self.gradient_clip_val = val
            self._apply_cfg()

    def log(self, name, value, *a, **k):
        self.logged[name] = value

    def train(self, mode=True):
        self.training = mode
        self.model.train(mode)
        # a schedule-free optimiser evaluates at the averaged point x and trains at y (hippie/optimizers.py:82-103)
        swap = getattr(getattr(self, "optimizer", None), "train" if mode else "eval", None)
        if swap is not None:
            swap()
        return self

    def eval(self):
        return self.train(False)

    def configure_optimizers(self):
        return self.optimizer

    def state_dict(self):
        return self.model.state_dict(prefix="model.")

    def load_state_dict(self, sd, strict=True):
        return self.model.load_state_dict(sd, strict=strict, prefix="model.")

    @staticmethod
    def _mean(values):
        if not values:
            return None
        if torch.is_tensor(values[0]):
            return float(torch.stack(values).double().mean())
        return sum(values) / len(values)

    def _record(self, store, loss):
        store.append(loss.item() if self.sync_every_step else loss.value)

    def on_validation_epoch_end(self):
        m = self._mean(self.val_loss)
        if m is not None:
            print(f"Average validation loss is {m:.2f}")
        self.last_val_mean = m
        self.val_loss = []

    def on_train_epoch_end(self):
        m = self._mean(self.train_loss)
        if m is not None:
            print(f"Average training loss is {m:.2f}")
        self.last_train_mean = m
        self.train_loss = []

    @staticmethod
    def _split_labels(labels):
        if labels.ndim == 2:                      # model.py:97-99: (class, source) = labels.unbind(1)
            c, s = labels.unbind(1)
            return s.contiguous(), c.contiguous()
        return labels, None


class hippieUnimodalEmbeddingModelCVAE(_TrainModule):
    """hippie/model.py:75-162."""

    def __init__(self, base_model, alpha_max=0.5, learning_rate=0.01, weight_decay=0.01, beta=1):
        self._setup(base_model, learning_rate, weight_decay, beta, alpha_max)

    def _step(self, batch, prefix, store):
        data, labels = batch
        src, cls = self._split_labels(labels)
        self.model.train(prefix == "train")
        eng = self.model.engine(data.shape[0], cls is not None)
        self.model._run_forward(eng, data, src, cls, None)
        sc = eng.io("scalars")                    # views: hold the values of the most recent step
        self.log(prefix + "_loss", sc[0])
        self.log(prefix + "_mse_loss", sc[1])
        self.log(prefix + "_kl_loss", sc[3])
        loss = _Loss(eng, use_graph=self.model.use_graph, dp_world=self.model.dp_world, dp_group=self.model.dp_group)
        self._record(store, loss)                 # loss.item() in the reference: one host sync per step
        self.optimizer.last_engine = eng
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, "train", self.train_loss)

    def validation_step(self, batch, batch_idx):
        return self._step(batch, "val", self.val_loss)

    def forward(self, batch):
        data, labels = batch
        src, cls = self._split_labels(labels)
        return self.model(data, source_labels=src, class_labels=cls)

    __call__ = forward

    def embed(self, batch):
        """`self(batch)[0]` without the decoder when the module is in eval mode (full forward otherwise)."""
        data, labels = batch
        src, cls = self._split_labels(labels)
        if self.model.training:
            return self.model(data, source_labels=src, class_labels=cls)[0]
        return self.model.encode_labels(data, src, cls)[0]


class MultiModalCVAETrainModule(_TrainModule):
    """hippie/model.py:434-533."""

    def __init__(self, base_model, alpha_max=0.5, learning_rate=0.01, weight_decay=0.01, beta=1, mod1_weight=1.0, mod2_weight=1.0):
        self._setup(base_model, learning_rate, weight_decay, beta, alpha_max, mod1_weight, mod2_weight)

    def _step(self, batch, prefix, store):
        data1, data2, labels = batch
        src, cls = self._split_labels(labels)
        self.model.train(prefix == "train")
        eng = self.model.engine(data1.shape[0], cls is not None)
        self.model._run_forward(eng, data1, src, cls, None, x2=data2)
        sc = eng.io("scalars")
        self.log(prefix + "_loss", sc[0])
        self.log(prefix + "_mse_loss1", sc[1])
        self.log(prefix + "_mse_loss2", sc[2])
        self.log(prefix + "_kl_loss", sc[3])
        loss = _Loss(eng, use_graph=self.model.use_graph, dp_world=self.model.dp_world, dp_group=self.model.dp_group)
        self._record(store, loss)
        self.optimizer.last_engine = eng
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, "train", self.train_loss)

    def validation_step(self, batch, batch_idx):
        return self._step(batch, "val", self.val_loss)

    def forward(self, batch):
        data1, data2, labels = batch
        src, cls = self._split_labels(labels)
        return self.model(data1, data2, source_labels=src, class_labels=cls)

    __call__ = forward
