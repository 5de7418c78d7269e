"""Schedule-Free AdamW on the MI355X engine — counterpart of hippie/optimizers.py:18-209.

Same constructor arguments, `step()` / `eval()` / `train()` / `state_dict()` surface and error behaviour as
the reference class; the arithmetic is the fused HP_OP_SF_SCHEDULE + HP_OP_ADAMW_SF launch pair in
libhippie_hip.so (z in the first-moment arena, exp_avg_sq in the second-moment arena, k / lr_max /
weight_sum on the device so the step is hipGraph-replayable).  There is no CPU path.

    opt = AdamWScheduleFree(model.parameters(), lr=1e-3, weight_decay=0.01, warmup_steps=100)
    module.optimizer = opt            # _TrainModule.train()/eval() then switch y <-> x for validation
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import replace


from . import program as P


class ParamHandle:
    """What `model.parameters()` returns here: parameters live in one flat HBM arena, so the optimiser gets the
    owner instead of a tensor list."""

    def __init__(self, net):
        self.net = net


class AdamWScheduleFree:
    def __init__(self, params, lr=0.0025, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_steps=0, r=0.0,
                 weight_lr_power=2.0, foreach=None):
        if not isinstance(params, ParamHandle):
            raise TypeError("AdamWScheduleFree(params=...): pass model.parameters() of a hippie_amd model")
        self.net = params.net
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, r=r, warmup_steps=warmup_steps,
                             weight_lr_power=weight_lr_power, weight_decay=weight_decay, foreach=foreach)
        self.train_mode = True
        self.last_engine = None
        self._apply()

    def _apply(self):
        d = self.defaults
        self.net.configure_training(replace(self.net._train_cfg, optimizer="schedulefree", lr=d["lr"], beta1=d["betas"][0],
                                            beta2=d["betas"][1], adam_eps=d["eps"], weight_decay=d["weight_decay"],
                                            warmup_steps=int(d["warmup_steps"]), sf_r=d["r"],
                                            sf_weight_lr_power=d["weight_lr_power"]))

    def _engine(self):
        eng = self.last_engine
        if eng is None or eng.train_cfg.optimizer != "schedulefree" or eng is not self.net._engines.get((eng.B, eng.with_class)):
            eng = self.net._any_engine()          # re-lowered since (e.g. gradient clipping changed)
        return eng

    # -- the reference's surface ------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        pass

    def eval(self):
        if self.train_mode:
            eng = self._engine()
            if eng.adam_step > 0:                 # "if 'z' in state" (:88)
                eng.optimizer_swap(True)
            self.train_mode = False

    def train(self):
        if not self.train_mode:
            eng = self._engine()
            if eng.adam_step > 0:
                eng.optimizer_swap(False)
            self.train_mode = True

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._engine()
        if not self.train_mode:
            # like the reference (:126-143), the schedule scalars have already advanced when this raises
            first, count = eng.plan.ops.segments["opt"]
            for k in range(first, first + count):
                if int(eng.ops[k]["op"]) == P.SF_SCHEDULE:
                    eng.prog.run(k, 1, eng._stream())
            raise Exception("Not in train mode!")
        eng.optimizer_step()
        return loss

    @property
    def param_groups(self):
        eng = self._engine()
        st = eng.io("sf_state").tolist()
        k = eng.adam_step
        g = dict(self.defaults)
        g.update(k=k, train_mode=self.train_mode, weight_sum=st[1], lr_max=st[0] if (k > 0 or st[0] > 0) else -1.0)
        return [g]

    def state_dict(self):
        eng = self._engine()
        state = OrderedDict()
        if eng.adam_step > 0:
            for i, key in enumerate(eng.plan.params):
                state[i] = dict(z=eng.param_view(key, eng.m).contiguous().clone(),
                                exp_avg_sq=eng.param_view(key, eng.v).contiguous().clone())
        n = len(eng.plan.params)
        return dict(state=state, param_groups=[dict(self.param_groups[0], params=list(range(n)))],
                    param_names=list(eng.plan.params))

    def load_state_dict(self, sd):
        eng = self._engine()
        names = sd.get("param_names", list(eng.plan.params))
        for i, key in enumerate(names):
            st = sd["state"].get(i)
            if st is None or key not in eng.plan.params or tuple(st["z"].shape) != tuple(eng.plan.params[key].shape):
                continue
            eng.param_view(key, eng.m).copy_(st["z"].to(eng.device))
            eng.param_view(key, eng.v).copy_(st["exp_avg_sq"].to(eng.device))
        g = sd["param_groups"][0]
        eng.io("adam_step").fill_(int(g["k"]))
        s = eng.io("sf_state")
        s[0], s[1] = max(float(g["lr_max"]), 0.0), float(g["weight_sum"])
        self.train_mode = bool(g.get("train_mode", True))
