"""Steady-state cost of every launch of one model's training step, measured op by op: each record of the lowered program
repeated back to back in a captured graph on the engine's own arenas (no per-launch event floor).  The sum is what the
step would cost as a purely serial chain; the table shows where the time of the small ops goes.
python tools/micro/op_chain_times.py [wave|time]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from hippie_amd import planner, program as P          # noqa: E402
from hippie_amd.engine import Engine                   # noqa: E402
from hippie_amd.program import DeviceProgram           # noqa: E402

REP = 30
which = sys.argv[1] if len(sys.argv) > 1 else "time"
L, clip = (50, 0.0) if which == "wave" else (100, 1.0)
eng = Engine(planner.ModelCfg(kind="unimodal", z_dim=10, output_size=L), 512, planner.TrainCfg(lr=1e-3, clip=clip))
g = torch.Generator(device="cpu").manual_seed(0)
eng.set_inputs(torch.randn(512, 1, L, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
eng.forward(True, False); eng.backward(False); eng.optimizer_step(False)
torch.cuda.synchronize()
arenas = [eng.ws, eng.params, eng.grads, eng.bufs, eng.m, eng.v]
bases, sizes = [a.data_ptr() for a in arenas], [a.numel() * a.element_size() for a in arenas]
ops, notes, segs = eng.ops, eng.plan.ops.notes, eng.plan.ops.segments
names = {getattr(P, n): n for n in dir(P) if n.isupper() and isinstance(getattr(P, n), int) and n not in ("NULL",)}
opname = lambda r: {v: k for k, v in P.__dict__.items() if k.isupper() and isinstance(v, int)}.get(int(r["op"]), str(int(r["op"])))
OPN = {}
import re
hdr = open("include/hippie_hip.h").read()
for n, v in re.findall(r"HP_OP_([A-Z_0-9]+) = (\d+)", hdr):
    OPN[int(v)] = n
stream = torch.cuda.current_stream().cuda_stream


def time_records(recs):
    """recs: the records of ONE launch (a lone record, or members + their PAIR / WGRAD_GROUP record)"""
    n = len(recs)
    out = []
    for rep in range(REP):
        for j, r in enumerate(recs):
            r = r.copy()
            opc = int(r["op"])
            if opc == P.PAIR:
                r["i"][0], r["i"][1] = rep * n + (int(recs[j]["i"][0]) - base), rep * n + (int(recs[j]["i"][1]) - base)
            elif opc in (P.WGRAD_GROUP, P.HEADS):
                r["i"][0] = rep * n + (int(recs[j]["i"][0]) - base)
            out.append(r)
    prog = DeviceProgram(np.array(out, dtype=P.OP_DTYPE), bases, sizes)
    seg = prog.capture(0, len(out))
    prog.replay(seg, stream)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.replay(seg, stream); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    prog.close()
    return best


import os
only = set(filter(None, os.environ.get("HIPPIE_ONLY_OP", "").split(",")))       # e.g. WGRAD_GROUP: time just those launches
rows = []
for seg in ("fwd_train", "bwd", "opt"):
    first, count = segs[seg]
    k = first
    member_of = {}
    for gidx in range(first, first + count):
        r = ops[gidx]
        if int(r["op"]) == P.PAIR:
            member_of[int(r["i"][0])] = gidx; member_of[int(r["i"][1])] = gidx
        elif int(r["op"]) == P.WGRAD_GROUP:
            for q in range(int(r["i"][0]), int(r["i"][0]) + int(r["i"][1])):
                member_of[q] = gidx
    done = set()
    for gidx in range(first, first + count):
        r = ops[gidx]
        if int(r["flags"]) & P.FLAG_MEMBER:
            continue
        opc = int(r["op"])
        if only and OPN.get(opc) not in only:
            continue
        nchain = (int(r["flags"]) >> P.FLAG_GROUP_SHIFT) & P.FLAG_GROUP_MASK
        if opc == P.PAIR:
            mem = sorted([int(r["i"][0]), int(r["i"][1])])
        elif opc in (P.WGRAD_GROUP, P.HEADS):
            mem = list(range(int(r["i"][0]), int(r["i"][0]) + int(r["i"][1])))
        elif nchain:
            mem = list(range(gidx - nchain, gidx))          # a chained / grouped launch: the preceding member records + this one
        else:
            mem = []
        if mem:
            base = mem[0]
            # members are contiguous and the group record follows (planner layout); rebase indices per repetition
            recs = [ops[q] for q in mem] + [r]
            us = time_records(recs)
        else:
            base = gidx
            us = time_records([r])
        rows.append((seg, OPN.get(opc, str(opc)), notes[gidx], us))
        print(f"{seg:9s} {OPN.get(opc, str(opc)):16s} {us:8.2f} us  {notes[gidx][:90]}", flush=True)
tot = sum(r[3] for r in rows)
print(f"TOTAL {len(rows)} launches, {tot / 1e3:.3f} ms as a serial chain ({which} model, batch 512)")
kinds = {}
for seg, nm, note, us in rows:
    d = kinds.setdefault(nm, [0, 0.0]); d[0] += 1; d[1] += us
for nm, (n, us) in sorted(kinds.items(), key=lambda x: -x[1][1]):
    print(f"  {nm:16s} {n:3d} launches {us:9.1f} us  {100 * us / tot:5.1f} %  avg {us / n:6.2f} us")
