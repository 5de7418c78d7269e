"""Diagnostic (GPU box): run one lowered program on the MI355X and in the numpy interpreter from the
same state, then walk the ops in order and report where outputs start to deviate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hippie_amd import planner, program as P
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O, interp
from tests import helpers as H

OUT_SLOTS = {P.CONV_TAPS: [(2, "MN")], P.SLAB_REDUCE: [(1, "n0")], P.BN_APPLY: [(1, "MC")], P.BN_BWD_REDUCE: [(3, "MC")],
             P.BN_BWD_APPLY: [(5, "MC")], P.STEM_FWD: [(2, "stem")], P.STEM_WGRAD: [(2, "192")], P.POOL_FWD: [(1, "BC")],
             P.POOL_BWD: [(1, "BLC")], P.REPEAT_FWD: [(1, "BLC")], P.REPEAT_BWD: [(2, "BC")], P.LINEAR_FWD: [(3, "lin_y")],
             P.LINEAR_BWD_X: [(2, "lin_dx")], P.LINEAR_BWD_W: [(2, "lin_dw")], P.TAIL_FWD: [(3, "tail")], P.TAIL_BWD_X: [(2, "BLC")],
             P.TAIL_BWD_W: [(2, "192")], P.REPARAM_KL_FWD: [(2, "Bz")], P.REPARAM_KL_BWD: [(3, "B2z")], P.MSE_FWD_BWD: [(2, "n0")]}

def count(r, kind):
    i = r["i"]
    return {"MN": i[0]*i[1], "n0": i[0], "MC": i[0]*i[1], "stem": i[0]*i[2]*i[3], "192": 192, "BC": i[0]*i[2], "BLC": i[0]*i[1]*i[2],
            "lin_y": (i[0]-1)*i[4]+i[1], "lin_dx": (i[0]-1)*i[4]+i[2], "lin_dw": i[1]*i[2], "tail": i[0]*2*i[1], "Bz": i[0]*i[1], "B2z": i[0]*2*i[1]}[kind]

z, L, B, salt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = planner.ModelCfg(kind="unimodal", z_dim=z, output_size=L)
eng = Engine(cfg, B, planner.TrainCfg(lr=1e-3))
om = O.OracleModel("unimodal", z, L, salt=salt)
eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt)
eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
plan, ops = eng.plan, eng.ops
A = H.make_arenas(plan)
H.load_state(plan, A, om.state)
for nme, v in (("x", x), ("src", src), ("cls", cls), ("eps", eps)): H.set_io(plan, A, nme, v.numpy())
eng.forward(True); eng.backward(); torch.cuda.synchronize()
segs = [plan.ops.segments[s] for s in ("fwd_train", "bwd")]
for s, c in segs: interp.run(ops, A, s, c)
ws = eng.ws.cpu().numpy(); gr = eng.grads.cpu().numpy().view(np.uint8)
mem = {P.WS: ws, P.GRAD: gr}
rows = []
for s, c in segs:
    for k in range(s, s + c):
        r = ops[k]; op = int(r["op"])
        for slot, kind in OUT_SLOTS.get(op, []):
            ref = int(r["buf"][slot]); sp, off = ref >> 56, ref & ((1 << 56) - 1)
            if sp not in mem: continue
            n = int(count(r, kind))
            g = mem[sp][off: off + 4*n].view(np.float32).astype(np.float64)
            cpu = A.mem[sp][off: off + 4*n].view(np.float32).astype(np.float64)
            sc = max(np.abs(cpu).max(), 1e-30)
            rows.append((k, P.OP_NAMES[op], plan.ops.notes[k], np.abs(g - cpu).max() / sc, sc))
print("%d ops compared" % len(rows))
thr = 0
for k, nm, note, e, sc in rows:
    if e > thr * 2 or e > 1e-4:
        print(f"{k:4d} {nm:16s} {note:44s} rel {e:.2e} scale {sc:.2e}")
        thr = max(thr, e)
