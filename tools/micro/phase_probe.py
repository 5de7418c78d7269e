"""Does the relative PHASE of the two model streams matter?  Free-running streams (the bench) drift through every alignment of
forward / backward chain / weight-gradient group / optimiser.  Here the alignment is forced with events between the segment graphs:
mode "free"      no cross-stream waits (whole-step graphs, the bench's schedule)
mode "free3"     the same with three graphs per model-step (what the forced modes cost by themselves)
mode "lock"      both models start step k together
mode "B@Afwd"    the time model's step k starts when the wave model's forward k is done; wave step k+1 waits for the time model's forward k
mode "B@Abwd"    ... when the wave model's backward k is done; wave step k+1 waits for the time model's backward k
python tools/micro/phase_probe.py"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, ".")
from hippie_amd import planner          # noqa: E402
from hippie_amd.engine import Engine     # noqa: E402

g = torch.Generator(device="cpu").manual_seed(0)
engs = []
for L, clip in ((50, 0.0), (100, 1.0)):
    e = Engine(planner.ModelCfg("unimodal", 10, L), 512, planner.TrainCfg(lr=1e-4, clip=clip))
    e.set_inputs(torch.randn(512, 1, L, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
    e.train_step(True)
    e.forward(True, True), e.backward(True), e.optimizer_step(True)
    engs.append(e)
A, B = engs
torch.cuda.synchronize()
pool = [torch.cuda.Stream() for _ in range(6)]
sA, sB = pool[2], pool[3]
REP = 40


def seg3(e, s, after_fwd=None, after_bwd=None):
    with torch.cuda.stream(s):
        e.forward(True, True)
        if after_fwd is not None:
            after_fwd.record(s)
        e.backward(True)
        if after_bwd is not None:
            after_bwd.record(s)
        e.optimizer_step(True)


def run(mode):
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(cur)
    sA.wait_stream(cur), sB.wait_stream(cur)
    prevB = None
    for _ in range(REP):
        if mode == "free":
            with torch.cuda.stream(sA):
                A.train_step(True)
            with torch.cuda.stream(sB):
                B.train_step(True)
        elif mode == "free3":
            seg3(A, sA), seg3(B, sB)
        elif mode == "lock":
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            ea.record(sA), eb.record(sB)
            sA.wait_event(eb), sB.wait_event(ea)
            seg3(A, sA), seg3(B, sB)
        else:
            at_fwd = mode == "B@Afwd"
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            if prevB is not None:
                sA.wait_event(prevB)
            seg3(A, sA, after_fwd=ea if at_fwd else None, after_bwd=None if at_fwd else ea)
            sB.wait_event(ea)
            seg3(B, sB, after_fwd=eb if at_fwd else None, after_bwd=None if at_fwd else eb)
            prevB = eb
    cur.wait_stream(sA), cur.wait_stream(sB)
    t1.record(cur)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / REP


best = None
REP = 6
for i in range(6):
    for j in range(6):
        if i != j:
            sA, sB = pool[i], pool[j]
            run("free")
            t = run("free")
            if best is None or t < best[0]:
                best = (t, i, j)
            print(f"pair ({i},{j}) {t:6.0f}", end=";  ", flush=True)
print("\nbest pair", best)
sA, sB = pool[best[1]], pool[best[2]]
REP = 40
for mode in ("free", "free3", "lock", "B@Afwd", "B@Abwd", "free"):
    run(mode)
    print(f"{mode:8s} {min(run(mode) for _ in range(3)):7.0f} us per pair-step", flush=True)
