"""Where does a CONV_TAPS launch spend its time?  One conv op repeated back to back in a captured graph, swept over
the number of K-slices (taps x K / 32) and the epilogue flags: the slope is the cost of a K-step, the intercept the
fixed cost of a launch (prologue, epilogue, drain).  Run on the GPU box: python tools/micro/conv_sweep.py"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref, TapMap   # noqa: E402

REP = 200


def time_op(tm, flags, ntapslab, reps=REP):
    nb = tm.M // tm.Lout
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    a = put(nb * tm.Lin * tm.K * 4)
    w = put(ntapslab * tm.N * tm.K * 4)
    out = put(tm.out_rows * tm.N * 4)
    bv = put(tm.N * 4)
    st = put(P.stat_repl(tm.N) * 2 * tm.N * 8)
    ol = P.OpList()
    for _ in range(reps):
        ol.add(P.CONV_TAPS, flags, tm.conv_ints(), (), [a, w, out, bv, st, None, None, None, None, None, None, None])
    recs = ol.array()
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev[: a.offset + nb * tm.Lin * tm.K * 4].view(torch.float32).normal_()
    sizes = [dev.numel(), 4, 4, 4, 4, 4]
    prog = P.DeviceProgram(recs, [dev.data_ptr()] + [0] * 5, sizes)
    seg = prog.capture(0, reps)
    s = torch.cuda.current_stream().cuda_stream
    prog.replay(seg, s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prog.replay(seg, s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    prog.close()
    return best


def time_null(reps=REP):
    """the floor: a chain of dependent launches that do (almost) nothing — HP_OP_ZERO of 16 bytes"""
    ol = P.OpList()
    for _ in range(reps):
        ol.add(P.ZERO, 0, [16, 0], (), [Ref(P.WS, 0)])
    dev = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    prog = P.DeviceProgram(ol.array(), [dev.data_ptr()] + [0] * 5, [dev.numel(), 4, 4, 4, 4, 4])
    seg = prog.capture(0, reps)
    s = torch.cuda.current_stream().cuda_stream
    prog.replay(seg, s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prog.replay(seg, s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    prog.close()
    return best


def main():
    print(f"chain of dependent 16-byte ZERO launches in a graph: {time_null():.2f} us per launch", flush=True)
    for nt, K in ((1, 32), (1, 64)):
        tm = TapMap(512 * 4, 512, K, 4, 4, 4, 1, 0, [(0, 0)] * nt)
        print(f"conv M=2048 N=512 K={K} taps={nt} ({nt * K // 32} K-steps): {time_op(tm, 0, 1):.2f} us", flush=True)
    shapes = {
        "L4  M=2048  N=512 K=512": (512, 4, 512, 512),
        "L3  M=3584  N=256 K=256": (512, 7, 256, 256),
        "L2  M=6656  N=128 K=128": (512, 13, 128, 128),
        "L1  M=12800 N=64  K=64 ": (512, 25, 64, 64),
        "L1x2 M=25600 N=64 K=64 ": (1024, 25, 64, 64),
    }
    variants = {"plain": 0, "stats": P.CONV_STATS, "kn": P.CONV_W_KN, "kn+stats": P.CONV_W_KN | P.CONV_STATS}
    for name, (B, L, N, K) in shapes.items():
        for vn, fl in variants.items():
            row = []
            for nt in (1, 2, 3, 6):
                taps = [((t % 3) - 1, t % 3) for t in range(nt)]
                tm = TapMap(B * L, N, K, L, L, L, 1, 0, taps)
                us = time_op(tm, fl, 3)
                row.append((nt * K // 32, us))
            (s0, t0), (s1, t1) = row[0], row[-1]
            slope = (t1 - t0) / (s1 - s0)
            print(f"{name} {vn:9s} " + " ".join(f"{s:3d} steps {t:6.2f} us" for s, t in row) +
                  f" | {slope * 1e3:6.1f} ns/step, fixed {t0 - slope * s0:5.2f} us", flush=True)


if __name__ == "__main__":
    main()
