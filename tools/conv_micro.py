"""Micro-benchmark of conv_taps_kernel on a few shapes (GPU box): median HIP-event time over back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hippie_amd import program as P
from hippie_amd.program import DeviceProgram, OpList, Ref, TapMap


def bench(M, N, K, L, w_kn=False, dbg=0, reps=30, stats=False):
    tm = TapMap(M, N, K, L, L, L, 1, 0, [(t - 1, t) for t in range(3)])
    a = torch.randn(M * K, device="cuda")
    w = torch.randn(3 * N * K, device="cuda") * 0.05
    out = torch.zeros(M * N, device="cuda")
    st = torch.zeros(2 * 16 * 2 * N, device="cuda")          # double[16][2][N] as floats
    ws = torch.cat([a, w, out, st]).contiguous()
    ra, rw, ro = Ref(P.WS, 0), Ref(P.WS, 4 * M * K), Ref(P.WS, 4 * (M * K + 3 * N * K))
    rs = Ref(P.WS, 4 * (M * K + 3 * N * K + M * N))
    ol = OpList()
    for _ in range(reps):
        ol.add(P.CONV_TAPS, (P.CONV_W_KN if w_kn else 0) | (P.CONV_STATS if stats else 0), tm.conv_ints(), (),
               [ra, rw, ro, None, rs if stats else None])
    dummy = torch.zeros(16, device="cuda")
    prog = DeviceProgram(ol.array(), [ws.data_ptr()] + [dummy.data_ptr()] * 5, [ws.numel() * 4] + [64] * 5)
    prog.profile(0, reps)
    ms = prog.profile(0, reps, torch.cuda.current_stream().cuda_stream)
    us = float(np.median(ms)) * 1e3
    return us, 2.0 * M * N * K * 3 / (us * 1e-6) / 1e12


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stats":
        for (M, N, K, L) in ((25600, 64, 64, 50), (12800, 128, 128, 25), (6656, 256, 256, 13), (3584, 512, 512, 7), (2048, 512, 512, 4)):
            for st in (False, True):
                us, tf = bench(M, N, K, L, False, stats=st)
                print(f"M={M} N={N} K={K} stats={int(st)}: {us:7.1f} us {tf:6.1f} TFLOP/s", flush=True)
        sys.exit(0)
    for (M, N, K, L) in ((2048, 512, 512, 4), (4096, 256, 256, 8), (16384, 64, 64, 32), (16384, 512, 512, 4)):
        for kn in (False, True):
            us, tf = bench(M, N, K, L, kn)
            print(f"M={M} N={N} K={K} [k][n]-weights={int(kn)}: {us:7.1f} us {tf:6.1f} TFLOP/s")
