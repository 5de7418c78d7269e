"""Free-running streams: which model finishes its K steps first, and does a higher stream priority for the longer chain (the time
model) balance the two?  T = max(finish) is what the bench measures.
python tools/micro/balance_probe.py"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, ".")
from hippie_amd import planner, streams          # noqa: E402
from hippie_amd.engine import Engine              # noqa: E402

g = torch.Generator(device="cpu").manual_seed(0)
engs = []
for L, clip in ((50, 0.0), (100, 1.0)):
    e = Engine(planner.ModelCfg("unimodal", 10, L), 512, planner.TrainCfg(lr=1e-4, clip=clip))
    e.set_inputs(torch.randn(512, 1, L, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
    e.train_step(True)
    engs.append(e)
torch.cuda.synchronize()
K = 60


def run(sA, sB):
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    t0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    t0.record(cur)
    sA.wait_stream(cur), sB.wait_stream(cur)
    for _ in range(K):
        for e, s in zip(engs, (sA, sB)):
            with torch.cuda.stream(s):
                e.train_step(True)
    ends[0].record(sA), ends[1].record(sB)
    torch.cuda.synchronize()
    return [t0.elapsed_time(x) * 1e3 / K for x in ends]


def best_pair(pool_a, pool_b):
    best = None
    for a in pool_a:
        for b in pool_b:
            if a is b:
                continue
            t = max(run(a, b))
            if best is None or t < best[0]:
                best = (t, a, b)
    return best


lo = [torch.cuda.Stream() for _ in range(5)]
hi = [torch.cuda.Stream(priority=-1) for _ in range(5)]
for name, pa, pb in (("both normal", lo, lo), ("time model high priority", lo, hi), ("wave model high priority", hi, lo), ("both high", hi, hi)):
    t, a, b = best_pair(pa, pb)
    fin = min((run(a, b) for _ in range(3)), key=max)
    print(f"{name:28s} wave finishes at {fin[0]:7.1f} us/step, time at {fin[1]:7.1f} us/step -> {max(fin):7.1f} us per pair-step", flush=True)
