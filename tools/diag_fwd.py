"""Diagnostic (GPU box): per-block forward accuracy and leaky-ReLU sign flips, engine vs torch-f32, both
measured against the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hippie_amd import planner, program as P
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O
torch.set_num_threads(8)
z, L, B, salt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = planner.ModelCfg(kind="unimodal", z_dim=z, output_size=L)
eng = Engine(cfg, B, planner.TrainCfg(lr=1e-3))
oms = [O.OracleModel("unimodal", z, L, salt=salt, dtype=dt) for dt in (torch.float32, torch.float64)]
eng.load_state_dict({k: v.detach() for k, v in oms[0].state.items()})
x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt)
eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
eng.forward(True); torch.cuda.synchronize()
taps = [{}, {}]
with torch.no_grad():
    oms[0].forward((x, src, None), eps, True, taps=taps[0])
    oms[1].forward((x.double(), src, None), eps.double(), True, taps=taps[1])
plan, ops = eng.plan, eng.ops
s, c = plan.ops.segments["fwd_train"]
print(f"{'block':34s} {'mine/f64':>10s} {'t32/f64':>10s} {'flips mine':>10s} {'flips t32':>10s}  n")
for k in range(s, s + c):
    r = ops[k]
    if int(r["op"]) != P.BN_APPLY or int(r["i"][2]) == 0: continue
    note = plan.ops.notes[k]
    tap = note.rsplit(".", 1)[0] + ".out"
    if tap not in taps[1]: continue
    M, C = int(r["i"][0]), int(r["i"][1])
    mine = eng.ws_f32(P.Ref(0, int(r["buf"][1]) & ((1 << 56) - 1)), M * C).cpu().numpy().reshape(B, M // B, C).transpose(0, 2, 1)
    t32, t64 = taps[0][tap].numpy(), taps[1][tap].numpy()
    sc = np.abs(t64).max()
    print(f"{tap:34s} {np.abs(mine - t64).max()/sc:10.2e} {np.abs(t32 - t64).max()/sc:10.2e} {int(((mine > 0) != (t64 > 0)).sum()):10d} {int(((t32 > 0) != (t64 > 0)).sum()):10d}  {mine.size}")
