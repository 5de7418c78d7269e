"""Why does the staged walk (bench.py's default path) drive the reconstruction error far below the conditional-mean floor that explicit
batches reach (tools/long_run_vs_oracle.py: mse ~0.011 with KL -> 0)?  Variants: staged with the fixed permutation, staged with a new
permutation every epoch, explicit random batches with torch eps.  Also: is the in-graph noise fresh every epoch?
python tools/debug/staged_longrun.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hippie_amd import planner
from hippie_amd.engine import Engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
N, B, z, L = 15631, 512, 10, 50
wave, _, labels = bench.synth_dataset(N, dev, lw=L)
spe = N // B


def make(resident):
    tc = planner.TrainCfg(lr=1e-3, weight_decay=0.01, beta=1.0, clip=0.0, resident_units=N if resident else 0)
    e = Engine(planner.ModelCfg("unimodal", z, L), B, tc, device=dev)
    from hippie_amd.model import reference_init_state
    e.load_state_dict(reference_init_state(e.cfg, torch.Generator().manual_seed(42)), strict=False)
    return e


g = torch.Generator().manual_seed(3)
for name in ("staged, fixed permutation", "staged, new permutation every epoch", "explicit random batches, torch noise"):
    e = make(name.startswith("staged"))
    if name.startswith("staged"):
        e.load_dataset(wave, labels, perm=torch.randperm(N, generator=g).to(dev), seed=77)
    eps_seen = {}
    for i in range(steps):
        if name.startswith("staged"):
            if "new permutation" in name and i % spe == 0:
                e.set_permutation(torch.randperm(N, generator=g).to(dev))
            e.train_step_staged(True)
            if i % spe == 3 and i // spe < 3:
                eps_seen[i // spe] = e.io("eps").detach().clone()
        else:
            idx = torch.randperm(N, generator=g)[:B].to(dev)
            e.set_inputs(wave[idx].view(B, 1, L), labels[idx], None, torch.randn(B, z, generator=g).to(dev))
            e.train_step(True)
        if i % (steps // 6) == 0 or i == steps - 1:
            sc = e.scalars()
            print(f"{name:40s} step {i:5d} loss {sc[0]:.6f} mse {sc[1]:.6f} kl {sc[3]:.6f}", flush=True)
    if eps_seen:
        a, b = eps_seen[0], eps_seen[1]
        print(f"   in-graph noise, batch 3 of epoch 0 vs epoch 1: mean {float(a.mean()):+.3f} std {float(a.std()):.3f}; identical across epochs: {bool(torch.equal(a, b))}; corr {float(torch.corrcoef(torch.stack([a.flatten(), b.flatten()]))[0, 1]):+.3f}")
