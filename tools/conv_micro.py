"""Micro-benchmark of conv_taps_kernel on a few shapes (GPU box): median HIP-event time over back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hippie_amd import program as P
from hippie_amd.program import DeviceProgram, OpList, Ref, TapMap


def bench(M, N, K, L, w_kn=False, dbg=0, reps=30):
    tm = TapMap(M, N, K, L, L, L, 1, 0, 0, [(t - 1, t) for t in range(3)])
    a = torch.randn(M * K, device="cuda")
    w = torch.randn(3 * N * K, device="cuda") * 0.05
    out = torch.zeros(M * N, device="cuda")
    ws = torch.cat([a, w, out]).contiguous()
    ra, rw, ro = Ref(P.WS, 0), Ref(P.WS, 4 * M * K), Ref(P.WS, 4 * (M * K + 3 * N * K))
    ol = OpList()
    for _ in range(reps):
        ol.add(P.CONV_TAPS, P.CONV_W_KN if w_kn else 0, tm.ints(), (), [ra, rw, ro, None, None])
    dummy = torch.zeros(16, device="cuda")
    prog = DeviceProgram(ol.array(), [ws.data_ptr()] + [dummy.data_ptr()] * 5, [ws.numel() * 4] + [64] * 5)
    prog.profile(0, reps)
    ms = prog.profile(0, reps, torch.cuda.current_stream().cuda_stream)
    us = float(np.median(ms)) * 1e3
    return us, 2.0 * M * N * K * 3 / (us * 1e-6) / 1e12


if __name__ == "__main__":
    for (M, N, K, L) in ((2048, 512, 512, 4), (4096, 256, 256, 8), (16384, 64, 64, 32), (16384, 512, 512, 4)):
        for kn in (False, True):
            us, tf = bench(M, N, K, L, kn)
            print(f"M={M} N={N} K={K} [k][n]-weights={int(kn)}: {us:7.1f} us {tf:6.1f} TFLOP/s")
