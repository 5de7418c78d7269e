"""Does a HIGH-priority HIP stream keep a latency-bound chain of small launches fast while a grid-filling kernel runs on another
stream?  Round 2's side-stream backward lost because every small kernel of the chain queued behind the weight-gradient group's
workgroups (profiles/r03_overlap_timeline.txt).  Probe: the time model's training forward (a hipGraph: ~80 launches, stem /
BatchNorm / head ops between the convs) on stream A, the grouped weight-gradient launch (3 workgroups per CU, ~300 us) looping on
stream B, for the stream-priority combinations.   python tools/micro/priority_probe.py"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (as bench.py: with the default 4, two default-priority streams can share a hardware queue and serialise)
import torch
sys.path.insert(0, ".")
from hippie_amd import planner, program as P          # noqa: E402
from hippie_amd.engine import Engine                   # noqa: E402

eng = Engine(planner.ModelCfg("unimodal", 10, 100), 512, planner.TrainCfg(lr=1e-4, clip=1.0))
g = torch.Generator(device="cpu").manual_seed(0)
eng.set_inputs(torch.randn(512, 1, 100, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
eng.forward(True, False); eng.backward(False); eng.optimizer_step(False)
torch.cuda.synchronize()
ops, segs = eng.ops, eng.plan.ops.segments
# the grouped 3-tap weight-gradient launch: members + group record
gi = next(k for k, r in enumerate(ops) if int(r["op"]) == P.WGRAD_GROUP and int(r["i"][2]) == 3)
first, count = int(ops[gi]["i"][0]), int(ops[gi]["i"][1]) + 1
wg = eng.prog.capture(first, count)
fwd = eng.prog.capture(*segs["fwd_train"])


def run(prio_chain, prio_group, with_group, reps=30):
    sa = torch.cuda.Stream(priority=prio_chain)
    sb = torch.cuda.Stream(priority=prio_group)
    torch.cuda.synchronize()
    if with_group:
        for _ in range(reps * 3):          # keeps stream B busy for the whole measurement (~300 us each)
            eng.prog.replay(wg, sb.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    time.sleep(0.002)                      # the group launches are in flight
    e0.record(sa)
    for _ in range(reps):
        eng.prog.replay(fwd, sa.cuda_stream)
    e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print(f"stream priority range (least, greatest): {lo} {hi}")
alone = run(0, 0, False)
print(f"forward graph alone: {alone:8.1f} us")
for pc, pg, name in ((0, 0, "equal priorities"), (hi, 0, "chain HIGH, group normal"), (hi, lo, "chain HIGH, group LOW"), (0, hi, "chain normal, group HIGH")):
    t = run(pc, pg, True)
    print(f"forward graph beside the looping wgrad group, {name:28s}: {t:8.1f} us  ({t / alone:.2f}x alone)")
