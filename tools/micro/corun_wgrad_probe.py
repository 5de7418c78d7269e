"""Upper bound for pairing a layer's weight-gradient GEMM with its input-gradient conv in ONE launch (both need the same dY, both are ready
together): the standalone WGRAD_TAPS launch of the layer (wave model's record, split so that it has a few hundred workgroups) on one
stream beside the same layer's dgrad conv (time model) on another; alone and together.
python tools/micro/corun_wgrad_probe.py"""
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
s0, s1 = streams.pick_concurrent_streams(engs, prioritise_longer=False)


def arenas(e):
    a = [e.ws, e.params, e.grads, e.bufs, e.m, e.v]
    return [t.data_ptr() for t in a], [t.numel() * t.element_size() for t in a]


def repeat_graph(e, recs):
    bases, sizes = arenas(e)
    prog = DeviceProgram(np.array(recs, dtype=P.OP_DTYPE), bases, sizes)
    gid = prog.capture(0, len(recs))
    return lambda st: prog.replay(gid, st)


def find(e, needle, seg, opc):
    first, count = e.plan.ops.segments[seg]
    for k in range(first, first + count):
        if needle in e.plan.ops.notes[k] and int(e.ops[k]["op"]) == opc:
            return e.ops[k]
    raise KeyError(needle)


def conv_graph(e, needle):
    r = find(e, needle, "bwd", P.CONV_TAPS).copy()
    r["flags"] = int(r["flags"]) & ~P.FLAG_MEMBER
    return repeat_graph(e, [r] * R)


def wgrad_graph(e, needle, nsplit):
    r = find(e, needle, "bwd", P.WGRAD_TAPS).copy()
    r["flags"] = int(r["flags"]) & ~P.FLAG_MEMBER
    M = int(r["i"][0])
    rows = -(-M // nsplit)
    rows = -(-rows // 32) * 32
    r["i"][22], r["i"][23] = -(-M // rows), rows
    tiles = -(-int(r["i"][1]) // 64) * -(-int(r["i"][2]) // 64)
    return repeat_graph(e, [r] * R), tiles * int(r["i"][22])


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
    return best / R


print(f"{'layer':26s} {'wgrad wgs':>9s} {'dgrad':>7s} {'wgrad':>7s} {'both':>7s}  both/sum  saved us")
for layer, splits in (("decoder.layer4.0.conv2", (2, 4, 8)), ("decoder.layer3.0.conv2", (8, 16)), ("decoder.layer2.0.conv2", (32, 64)),
                      ("decoder.layer1.0.conv2", (128, 256)), ("encoder.layer4.1.conv1", (4, 8)), ("encoder.layer2.1.conv1", (32, 64)),
                      ("encoder.layer1.1.conv1", (128, 256))):
    fc = conv_graph(engs[1], layer + " dgrad")
    fc(torch.cuda.current_stream().cuda_stream)
    tc = timed([(fc, s1)])
    for ns in splits:
        fw, wgs = wgrad_graph(engs[0], layer + " wgrad", ns)
        fw(torch.cuda.current_stream().cuda_stream)
        tw = timed([(fw, s0)])
        tb = timed([(fw, s0), (fc, s1)])
        print(f"{layer:26s} {wgs:9d} {tc:7.1f} {tw:7.1f} {tb:7.1f}  {tb / (tc + tw):8.2f}  {tc + tw - tb:7.1f}", flush=True)
