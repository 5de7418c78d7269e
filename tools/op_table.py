"""Per-op HIP-event times of one model's training step (eager, serial), with algorithmic bytes for the
BatchNorm family.  python tools/op_table.py [L] [B]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippie_amd import planner, program as P
from hippie_amd.engine import Engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = Engine(planner.ModelCfg("unimodal", 10, L), B, planner.TrainCfg(lr=1e-3, clip=1.0))
from bench import Pair
x = torch.randn(B, 1, L, device="cuda")
eng.set_inputs(x, torch.randint(1, 5, (B,), device="cuda"))
for _ in range(3):
    eng.train_step()
torch.cuda.synchronize()
rows = []
for seg in ("fwd_train", "bwd", "opt"):
    acc = None
    for rep in range(5):
        eng.forward(True) if seg != "fwd_train" else None
        if seg == "opt":
            eng.backward()
        r = eng.profile(seg)
        acc = r if acc is None else [(a[0], a[1], a[2] + b[2]) for a, b in zip(acc, r)]
    first, count = eng.plan.ops.segments[seg]
    for k, (name, note, ms) in enumerate(acc):
        rec = eng.ops[first + k]
        rows.append((seg, name, note, ms / 5 * 1e3, rec))
tot = sum(r[3] for r in rows)
print(f"total {tot:.1f} us")
for seg, name, note, us, rec in rows:
    extra = ""
    if name in ("BN_APPLY", "BN_BWD_REDUCE", "BN_BWD_APPLY"):
        M, C = int(rec["i"][0]), int(rec["i"][1])
        nb = sum(1 for b in rec["buf"] if int(b) != P.NULL)
        extra = f"M={M} C={C} nbuf={nb} tensorMB={M*C*4/1e6:.2f} flags={int(rec['flags'])}"
    if us > 0:
        print(f"{seg:9s} {name:14s} {us:8.1f} us  {note[:60]:60s} {extra}")
