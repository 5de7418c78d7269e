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
        self._pending_sd = None
        self._generation = 0

    # -- engine cache -----------------------------------------------------------------
    def engine(self, batch, with_class) -> Engine:
        key = (int(batch), bool(with_class))
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(self.cfg, key[0], self._train_cfg, with_class=key[1], device=self.device, share_params_from=self._root)
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
        self._train_cfg = train_cfg
        keep = self._root
        self._engines = {}
        if keep is not None:
            eng = Engine(self.cfg, keep.B, train_cfg, with_class=keep.with_class, device=self.device, share_params_from=keep)
            if reset_optimizer:
                eng.reset_optimizer_state()
            self._root = eng
            self._engines[(keep.B, keep.with_class)] = eng
        self._generation += 1

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
        eng.set_inputs(x, src, cls, eps, x2=x2)
        return eng.forward(training=self.training)


def _default_init(eng: Engine, seed=None):
    """torch's default initialisers (kaiming-uniform(a=sqrt 5) => U(+-1/sqrt(fan_in)) for weights and biases,
    BatchNorm 1/0, Embedding N(0,1)), drawn from the torch CPU generator like the reference's constructors."""
    g = None
    if seed is not None:
        g = torch.Generator().manual_seed(seed)
    sd = {}
    bn_prefixes = set(eng.plan.bn_keys)
    for k, info in eng.plan.params.items():
        shp = info.shape
        pre = k.rsplit(".", 1)[0]
        if k.endswith("embedding.weight"):
            v = torch.randn(shp, generator=g)
        elif pre in bn_prefixes:
            v = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        else:
            wshape = eng.plan.params[pre + ".weight"].shape
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            bound = 1.0 / fan_in ** 0.5
            v = (torch.rand(shp, generator=g) * 2 - 1) * bound
        sd[k] = v
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
        eng.set_inputs(data, source_labels, class_labels, eps=torch.zeros_like(eng.io("eps")))
        return eng.encode()


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

    def __init__(self, eng: Engine, slot=0):
        self.eng, self.slot = eng, slot
        self._backward_done = False

    def item(self):
        return float(self.eng.io("scalars")[self.slot])

    __float__ = item

    def detach(self):
        return self.eng.io("scalars")[self.slot].clone()

    def backward(self):
        if not self._backward_done:
            self.eng.backward()
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
        eng.optimizer_step()

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

    def on_validation_epoch_end(self):
        if self.val_loss:
            print(f"Average validation loss is {sum(self.val_loss) / len(self.val_loss):.2f}")
        self.val_loss = []

    def on_train_epoch_end(self):
        if self.train_loss:
            print(f"Average training loss is {sum(self.train_loss) / len(self.train_loss):.2f}")
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
        sc = eng.io("scalars")
        self.log(prefix + "_loss", sc[0])
        self.log(prefix + "_mse_loss", sc[1])
        self.log(prefix + "_kl_loss", sc[3])
        loss = _Loss(eng)
        store.append(loss.item())                 # loss.item() in the reference: one host sync per step
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
        loss = _Loss(eng)
        store.append(loss.item())
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
