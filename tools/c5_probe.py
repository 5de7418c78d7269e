"""One-GPU probe of BASELINE config 5's per-rank shape: multimodal cVAE, z=64, wave 256 + time 32, batch 8192."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippie_amd import planner
from hippie_amd.engine import Engine

B, z = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 64
eng = Engine(planner.ModelCfg("multimodal", z, 256, 32), B, planner.TrainCfg(lr=1e-3, clip=1.0))
print(f"workspace {eng.ws.numel()/1e9:.1f} GB, params {eng.plan.n_param_floats/1e6:.1f} M floats, {len(eng.plan.ops.recs)} ops", flush=True)
from hippie_amd.model import _default_init
_default_init(eng, seed=0)
x1, x2 = torch.randn(B, 1, 256, device="cuda"), torch.rand(B, 1, 32, device="cuda")
eng.set_inputs(x1, torch.randint(1, 5, (B,), device="cuda"), x2=x2)
for i in range(3):
    eng.train_step(use_graph=True)
torch.cuda.synchronize()
print("loss, mse1, mse2, kl:", eng.scalars(), flush=True)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    eng.set_inputs(x1, torch.randint(1, 5, (B,), device="cuda"), x2=x2)
    eng.train_step(use_graph=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 3.0 * eng.plan.flops_fwd
print(f"{dt*1e3:.1f} ms per step -> {B/dt:.0f} units/s, {fl/dt/1e12:.1f} TFLOP/s whole step; loss {eng.scalars()[0]:.4f}", flush=True)
