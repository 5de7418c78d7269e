"""Fixed cost of conv_taps_kernel: same M/N, contraction length swept (taps x K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hippie_amd import program as P
from hippie_amd.program import DeviceProgram, OpList, Ref, TapMap

def bench(M, N, K, L, ntaps, reps=30):
    taps = [(t - 1, t) for t in range(3)][:ntaps] if ntaps > 1 else [(0, 0)]
    tm = TapMap(M, N, K, L, L, L, 1, 0, taps)
    ws = torch.randn(M * K + 3 * N * K + M * N, device="cuda") * 0.05
    ra, rw, ro = Ref(P.WS, 0), Ref(P.WS, 4 * M * K), Ref(P.WS, 4 * (M * K + 3 * N * K))
    ol = OpList()
    for _ in range(reps):
        ol.add(P.CONV_TAPS, 0, tm.conv_ints(), (), [ra, rw, ro, None, None])
    dummy = torch.zeros(16, device="cuda")
    prog = DeviceProgram(ol.array(), [ws.data_ptr()] + [dummy.data_ptr()] * 5, [ws.numel() * 4] + [64] * 5)
    prog.profile(0, reps)
    ms = prog.profile(0, reps, torch.cuda.current_stream().cuda_stream)
    return float(np.median(ms)) * 1e3

for (M, N, L) in ((25600, 64, 50), (12800, 128, 25), (3584, 512, 7)):
    for K, nt in ((32, 1), (64, 1), (64, 3), (128, 3), (256, 3)):
        us = bench(M, N, K, L, nt)
        print(f"M={M} N={N} K={K} taps={nt} (steps {nt * K // 32:3d}): {us:6.1f} us", flush=True)
