"""Embedding-extraction rate (scripts/utils.py:get_embeddings path): eval forward vs encoder-only, wave + time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippie_amd import planner
from hippie_amd.engine import Engine

for B in (512, 4096):
    engs = [Engine(planner.ModelCfg("unimodal", 10, L), B) for L in (50, 100)]
    for e, L in zip(engs, (50, 100)):
        e.set_inputs(torch.randn(B, 1, L, device="cuda"), torch.randint(1, 5, (B,), device="cuda"))
    for name, fn in (("eval forward (enc, mu, logvar, dec)", lambda e: e.forward(False, True)), ("encoder only", lambda e: e.encode(True))):
        for _ in range(5):
            for e in engs:
                fn(e)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            for e in engs:
                fn(e)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:5d} {name:38s} {dt*1e3:7.3f} ms per batch (wave+time, one stream) -> {B/dt:10.0f} units/s", flush=True)
