"""Do the two model streams really run side by side?  tools/stream_overlap.py on the default bench trace: conv kernels of one model
spend only 12-20 % of their time beside a kernel of the other model — the streams mostly ALTERNATE.  Probe: the wave and the time
model's whole step graphs on every pair out of S freshly created streams (ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues,
which the firmware spreads over a few pipes): pair time against the two graphs alone and back to back.
python tools/micro/queue_pair_probe.py [n_streams]"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[2] if len(sys.argv) > 2 else "8")
import torch
sys.path.insert(0, ".")
from hippie_amd import planner          # noqa: E402
from hippie_amd.engine import Engine     # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
engs = []
g = torch.Generator(device="cpu").manual_seed(0)
for L, clip in ((50, 0.0), (100, 1.0)):
    e = Engine(planner.ModelCfg("unimodal", 10, L), 512, planner.TrainCfg(lr=1e-4, clip=clip))
    e.set_inputs(torch.randn(512, 1, L, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
    e.train_step(True)
    engs.append(e)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(S)]
REP = 20


def timed(assign):
    """assign: [(engine, stream), ...]; every engine replays its step graph REP times on its stream"""
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _, s in assign:
        s.wait_stream(cur)
    for _ in range(REP):
        for e, s in assign:
            with torch.cuda.stream(s):
                e.train_step(True)
    for _, s in assign:
        cur.wait_stream(s)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP


a0 = timed([(engs[0], streams[0])])
a1 = timed([(engs[1], streams[0])])
print(f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}  wave step alone {a0:.0f} us, time step alone {a1:.0f} us, back to back {a0 + a1:.0f} us")
print("pair time (us) for wave on stream i (row), time on stream j (column):")
print("      " + " ".join(f"{j:6d}" for j in range(S)))
for i in range(S):
    row = []
    for j in range(S):
        row.append(timed([(engs[0], streams[i]), (engs[1], streams[j])]))
    print(f"{i:4d}  " + " ".join(f"{v:6.0f}" for v in row))
