"""Cost structure of the weight-gradient kernel: one WGRAD_TAPS (atomic) op repeated in a captured graph, swept over
rows per split (32-row slices per block) and workgroups per CU.  python tools/micro/wgrad_sweep.py"""
import sys
import torch

sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref, TapMap   # noqa: E402

REP = 50


def time_op(tm, nsplit, rps, reps=REP, in_bn=False):
    nb = tm.M // tm.Lout
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    dy = put(tm.M * tm.N * 4)
    x = put(nb * tm.Lin * tm.K * 4)
    numel = len(tm.taps) * tm.N * tm.K
    g = put(numel * 4)
    coef = put(2 * tm.K * 4)
    ol = P.OpList()
    for _ in range(reps):
        ol.add(P.WGRAD_TAPS, 1 | (P.CONV_IN_BN if in_bn else 0), tm.ints() + [nsplit, rps, numel], [0.01], [dy, x, g, coef if in_bn else None])
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev[: g.offset].view(torch.float32).normal_()
    prog = P.DeviceProgram(ol.array(), [dev.data_ptr()] + [0] * 5, [dev.numel(), 4, 4, 4, 4, 4])
    seg = prog.capture(0, reps)
    s = torch.cuda.current_stream().cuda_stream
    prog.replay(seg, s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prog.replay(seg, s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    prog.close()
    return best


IN_BN = len(sys.argv) > 1 and sys.argv[1] == "in_bn"      # X is a raw BatchNorm input: the loader re-evaluates the activation


def main():
    print("weight-gradient operand X:", "raw BatchNorm input (HP_CONV_IN_BN)" if IN_BN else "stored tensor", flush=True)
    for name, (L, N, K) in {"L4 512x512": (4, 512, 512), "L3 256x256": (7, 256, 256), "L1 64x64": (25, 64, 64)}.items():
        tiles = (N // 64) * (K // 64)
        for wg_per_cu in (3,):
            nsplit = max(1, 256 * wg_per_cu // tiles)
            row = []
            for slices in (4, 8, 16, 32):
                rps = 32 * slices
                M = nsplit * rps
                B = -(-M // L)
                M = B * L
                ns = -(-M // rps)
                tm = TapMap(M, N, K, L, L, L, 1, 0, [(t - 1, t) for t in range(3)])
                us = time_op(tm, ns, rps, in_bn=IN_BN)
                row.append((slices, us, 2.0 * M * N * K * 3 / us * 1e-6))
            (s0, t0, _), (s1, t1, _) = row[0], row[-1]
            slope = (t1 - t0) / (s1 - s0)
            print(f"{name} {tiles * nsplit:5d} blocks ({wg_per_cu}/CU) " + " ".join(f"{s:2d} sl {t:7.2f} us {tf:5.1f} TF" for s, t, tf in row) +
                  f" | {slope * 1e3:7.1f} ns/slice, fixed {t0 - slope * s0:6.2f} us", flush=True)


if __name__ == "__main__":
    main()
