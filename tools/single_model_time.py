"""Wall time of ONE model's training step alone (graph replay vs eager), to compare with the per-op sums."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippie_amd import planner
from hippie_amd.engine import Engine

B = 512
for L in (50, 100):
    eng = Engine(planner.ModelCfg("unimodal", 10, L), B, planner.TrainCfg(lr=1e-3, clip=1.0 if L == 100 else 0.0))
    x = torch.randn(B, 1, L, device="cuda")
    eng.set_inputs(x, torch.randint(1, 5, (B,), device="cuda"))
    for use_graph in (False, True):
        for _ in range(10):
            eng.train_step(use_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            eng.train_step(use_graph)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"L={L} graph={use_graph}: {dt*1e3:.3f} ms/step", flush=True)
        for seg in ("fwd_train", "bwd", "opt"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                eng.run(seg, use_graph)
            torch.cuda.synchronize()
            print(f"    {seg}: {(time.perf_counter()-t0)/n*1e3:.3f} ms", flush=True)
