"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's Schedule-Free AdamW.

Follows hippie/optimizers.py:18-209 (vendored from Meta's schedule_free): one
parameter group, state {z, exp_avg_sq} per tensor and {k, weight_sum, lr_max}
per group.  Pinned by tests/golden/schedulefree_*.npz, which were produced by
the reference class itself (tests/golden/make_golden_optim.py).  Only tests/
may import this module; the product path never does.
"""
from __future__ import annotations

import math

import torch


class ScheduleFreeOracle:
    def __init__(self, params: dict, lr=0.0025, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, warmup_steps=0,
                 r=0.0, weight_lr_power=2.0):
        self.params = params                      # name -> tensor, updated in place (this is "y" in train mode)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.warmup_steps, self.r, self.weight_lr_power = warmup_steps, r, weight_lr_power
        self.k, self.weight_sum, self.lr_max = 0, 0.0, -1.0          # optimizers.py:66-79
        self.train_mode = True
        self.z, self.exp_avg_sq = {}, {}

    def eval(self):                               # optimizers.py:82-92: p <- x
        if self.train_mode:
            for n, p in self.params.items():
                if n in self.z:
                    p.lerp_(self.z[n], 1 - 1 / self.betas[0])
            self.train_mode = False

    def train(self):                              # optimizers.py:94-103: p <- y
        if not self.train_mode:
            for n, p in self.params.items():
                if n in self.z:
                    p.lerp_(self.z[n], 1 - self.betas[0])
            self.train_mode = True

    def schedule(self):
        """(lr_t, ckp1) of the coming step; advances lr_max and weight_sum (optimizers.py:126-143)."""
        k, (beta1, beta2) = self.k, self.betas
        sched = (k + 1) / self.warmup_steps if k < self.warmup_steps else 1.0
        lr_t = self.lr * sched * math.sqrt(1 - beta2 ** (k + 1))
        self.lr_max = max(lr_t, self.lr_max)
        weight = ((k + 1) ** self.r) * (self.lr_max ** self.weight_lr_power)
        self.weight_sum += weight
        ckp1 = weight / self.weight_sum if self.weight_sum != 0 else 0
        return lr_t, ckp1

    def step(self, grads: dict):
        lr_t, ckp1 = self.schedule()
        if not self.train_mode:
            raise Exception("Not in train mode!")
        beta1, beta2 = self.betas
        for n, y in self.params.items():
            g = grads.get(n)
            if g is None:
                continue
            if n not in self.z:
                self.z[n] = y.clone()
                self.exp_avg_sq[n] = torch.zeros_like(y)
            v, z = self.exp_avg_sq[n], self.z[n]
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            gn = g / (v.sqrt() + self.eps)
            if self.weight_decay != 0:
                gn = gn + self.weight_decay * y           # decay evaluated at y (:193-195)
            y.lerp_(z, ckp1)
            y.add_(gn, alpha=lr_t * (beta1 * (1 - ckp1) - 1))
            z.sub_(gn, alpha=lr_t)
        self.k += 1
