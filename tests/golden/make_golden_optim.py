"""Generate schedule-free AdamW golden vectors from the reference's own class.

Run in the build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_optim.py
Imports hippie/optimizers.py:AdamWScheduleFree, drives it on seeded tensors and stores inputs + every
intermediate state.  The reference never travels; only these arrays do.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_optimizers", "/root/reference/hippie/optimizers.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

CASES = {
    # name: (kwargs, foreach, steps, swap_after)   swap_after: eval()+train() round trip after that step
    "default": (dict(), True, 4, None),
    "warmup_decay": (dict(lr=1e-2, weight_decay=0.01, warmup_steps=3, r=0.5, weight_lr_power=2.0), True, 6, 2),
    "loop_path": (dict(lr=5e-3, betas=(0.8, 0.99), weight_decay=0.1, warmup_steps=2), False, 5, 3),
}
SHAPES = [(7, 5), (13,), (3, 4, 3), (1030,)]


def main():
    for name, (kw, foreach, steps, swap_after) in CASES.items():
        gen = torch.Generator().manual_seed(1234 + len(name))
        params = [torch.randn(s, generator=gen).mul_(0.3).requires_grad_(True) for s in SHAPES]
        opt = ref.AdamWScheduleFree(params, foreach=foreach, **kw)
        out = {"n_steps": np.int64(steps), "swap_after": np.int64(-1 if swap_after is None else swap_after)}
        for j, p in enumerate(params):
            out[f"p0_{j}"] = p.detach().numpy().copy()
        for t in range(steps):
            for j, p in enumerate(params):
                g = torch.randn(p.shape, generator=gen) * (0.5 + 0.1 * t)
                out[f"g{t}_{j}"] = g.numpy().copy()
                p.grad = g.clone()                    # the class normalises .grad in place
            opt.step()
            grp = opt.param_groups[0]
            out[f"group{t}"] = np.array([grp["k"], grp["weight_sum"], grp["lr_max"]], dtype=np.float64)
            for j, p in enumerate(params):
                out[f"y{t}_{j}"] = p.detach().numpy().copy()
                out[f"z{t}_{j}"] = opt.state[p]["z"].numpy().copy()
                out[f"v{t}_{j}"] = opt.state[p]["exp_avg_sq"].numpy().copy()
            if swap_after is not None and t == swap_after:
                opt.eval()
                for j, p in enumerate(params):
                    out[f"x{t}_{j}"] = p.detach().numpy().copy()
                try:
                    opt.step()
                    out["eval_step_raises"] = np.int64(0)
                except Exception as e:                 # "Not in train mode!"
                    out["eval_step_raises"] = np.int64(1)
                    out["eval_step_message"] = np.array(str(e))
                    # the reference has already advanced lr_max / weight_sum before raising (:126-143)
                    grp = opt.param_groups[0]
                    out[f"group_after_raise"] = np.array([grp["k"], grp["weight_sum"], grp["lr_max"]], dtype=np.float64)
                opt.train()
                for j, p in enumerate(params):
                    out[f"yback{t}_{j}"] = p.detach().numpy().copy()
        np.savez_compressed(os.path.join(HERE, f"schedulefree_{name}.npz"), **out)
        print(name, "ok", {k: v for k, v in opt.param_groups[0].items() if k != "params"})


if __name__ == "__main__":
    main()
