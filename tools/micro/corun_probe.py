"""How do two kernels of the two models share the GPU?  For pairs (X from the wave model, Y from the time model): X repeated R times
back to back in a graph on one stream, Y likewise on another (a stream pair that overlaps: hippie_amd/streams.py); alone and together.
T_both = max(tX, tY): free overlap;  = tX + tY: the two only alternate;  in between: they share the matrix pipe / workgroup slots.
python tools/micro/corun_probe.py"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
sys.path.insert(0, ".")
from hippie_amd import planner, program as P, streams          # noqa: E402
from hippie_amd.engine import Engine                             # noqa: E402
from hippie_amd.program import DeviceProgram                      # noqa: E402

R = 30
g = torch.Generator(device="cpu").manual_seed(0)
engs = []
for L, clip in ((50, 0.0), (100, 1.0)):
    e = Engine(planner.ModelCfg("unimodal", 10, L), 512, planner.TrainCfg(lr=1e-4, clip=clip))
    e.set_inputs(torch.randn(512, 1, L, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
    e.train_step(True)
    engs.append(e)
torch.cuda.synchronize()
s0, s1 = streams.pick_concurrent_streams(engs)


def unit_graph(e, needle, seg):
    """(replay function, description) for the launch whose note contains `needle` in segment `seg`"""
    ops, notes = e.ops, e.plan.ops.notes
    first, count = e.plan.ops.segments[seg]
    arenas = [e.ws, e.params, e.grads, e.bufs, e.m, e.v]
    bases, sizes = [a.data_ptr() for a in arenas], [a.numel() * a.element_size() for a in arenas]
    for k in range(first, first + count):
        r = ops[k]
        if needle not in notes[k] or int(r["flags"]) & P.FLAG_MEMBER:
            continue
        opc = int(r["op"])
        if opc == P.WGRAD_GROUP:
            i0, n = int(r["i"][0]), int(r["i"][1])
            gid = e.prog.capture(i0, k - i0 + 1)
            reps = 3
            return (lambda st: [e.prog.replay(gid, st) for _ in range(reps)]), f"{notes[k]} x{reps}", reps
        unit = [ops[int(r["i"][0])], ops[int(r["i"][1])], r] if opc == P.PAIR else [r]
        recs = []
        for rep in range(R):
            for u in unit:
                u = u.copy()
                if int(u["op"]) == P.PAIR:
                    u["i"][0], u["i"][1] = rep * 3, rep * 3 + 1
                recs.append(u)
        prog = DeviceProgram(np.array(recs, dtype=P.OP_DTYPE), bases, sizes)
        gid = prog.capture(0, len(recs))
        return (lambda st: prog.replay(gid, st)), notes[k], R
    raise KeyError(needle)


def timed(jobs):
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    best = 1e30
    for _ in range(4):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(cur)
        for _, s in jobs:
            s.wait_stream(cur)
        for fn, s in jobs:
            fn(s.cuda_stream)
        for _, s in jobs:
            cur.wait_stream(s)
        t1.record(cur)
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3)
    return best


CASES = [
    (("encoder.layer1.0.conv1", "fwd_train"), ("encoder.layer1.0.conv1", "fwd_train")),
    (("encoder.layer3.1.conv1", "fwd_train"), ("encoder.layer3.1.conv1", "fwd_train")),
    (("decoder.layer4.0.conv2", "fwd_train"), ("decoder.layer4.0.conv2", "fwd_train")),
    (("decoder.layer4.0.conv2", "fwd_train"), ("encoder.layer1.0.conv1", "fwd_train")),
    (("decoder.layer4.0.conv2", "fwd_train"), ("encoder_fc.0", "fwd_train")),
    (("decoder.layer4.0.conv2", "fwd_train"), ("decoder_fc.2 dX", "bwd")),
    (("encoder.layer1.0.conv1", "fwd_train"), ("encoder_fc.0", "fwd_train")),
    (("encoder_fc.0", "fwd_train"), ("encoder_fc.0", "fwd_train")),
    (("decoder.layer4.0.conv2 dgrad", "bwd"), ("decoder.layer4.0.conv2 dgrad", "bwd")),
    (("grouped wgrad x35", "bwd"), ("decoder.layer4.0.conv2", "fwd_train")),
    (("grouped wgrad x35", "bwd"), ("encoder.layer1.0.conv1", "fwd_train")),
    (("grouped wgrad x35", "bwd"), ("encoder_fc.0", "fwd_train")),
    (("grouped wgrad x35", "bwd"), ("grouped wgrad x35", "bwd")),
]
if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
    CASES = [c for c in CASES if "wgrad" in c[0][0]]
print(f"HIPPIE_WGRAD_PERSIST={os.environ.get('HIPPIE_WGRAD_PERSIST', '0')}")
print(f"{'X (wave model)':44s} {'Y (time model)':44s} {'tX':>8s} {'tY':>8s} {'both':>8s}  both/max  both/sum   (us per launch of X | Y alone)")
for (nx, sx), (ny, sy) in CASES:
    fx, dx, rx = unit_graph(engs[0], nx, sx)
    fy, dy, ry = unit_graph(engs[1], ny, sy)
    for f in (fx, fy):
        f(torch.cuda.current_stream().cuda_stream)
    tx, ty = timed([(fx, s0)]), timed([(fy, s1)])
    tb = timed([(fx, s0), (fy, s1)])
    print(f"{dx[:44]:44s} {dy[:44]:44s} {tx:8.1f} {ty:8.1f} {tb:8.1f}  {tb / max(tx, ty):8.2f}  {tb / (tx + ty):8.2f}   ({tx / rx:.1f} | {ty / ry:.1f})", flush=True)
