"""Three stream priority levels?  torch.cuda.Stream offers two (0, -1); HIP reports its own range.  Streams made with
hipStreamCreateWithPriority through ctypes, wrapped as torch.cuda.ExternalStream; pair-step time for (wave, time) priority combinations.
python tools/micro/priority3_probe.py"""
import ctypes
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, ".")
from hippie_amd import planner          # noqa: E402
from hippie_amd.engine import Engine     # noqa: E402

torch.cuda.init()
hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        pass
lo, hi = ctypes.c_int(), ctypes.c_int()
assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
print(f"hipDeviceGetStreamPriorityRange: least {lo.value}, greatest {hi.value}")


def make(prio, n=5):
    out = []
    for _ in range(n):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, prio) == 0      # 1 = hipStreamNonBlocking
        out.append(torch.cuda.ExternalStream(s.value))
    return out


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
    ring = [[None, None], [None, None]]
    t0.record(cur)
    sA.wait_stream(cur), sB.wait_stream(cur)
    for i in range(K):
        for k in range(2):
            if ring[k][i % 2] is not None:
                ring[k][i % 2].synchronize()
        for k, (e, s) in enumerate(zip(engs, (sA, sB))):
            with torch.cuda.stream(s):
                e.train_step(True)
            ring[k][i % 2] = torch.cuda.Event()
            ring[k][i % 2].record(s)
    ends[0].record(sA), ends[1].record(sB)
    torch.cuda.synchronize()
    return [t0.elapsed_time(x) * 1e3 / K for x in ends]


levels = sorted({lo.value, 0, hi.value})
pools = {p: make(p) for p in levels}
for pa in levels:
    for pb in levels:
        best = None
        for a in pools[pa]:
            for b in pools[pb]:
                if a is b:
                    continue
                t = max(run(a, b))
                if best is None or t < best[0]:
                    best = (t, a, b)
        fin = min((run(best[1], best[2]) for _ in range(3)), key=max)
        print(f"wave priority {pa:2d}, time priority {pb:2d}: wave ends {fin[0]:7.1f}, time ends {fin[1]:7.1f} -> {max(fin):7.1f} us per pair-step", flush=True)
