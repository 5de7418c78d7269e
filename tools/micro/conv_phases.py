"""Phase timestamps (100 MHz wall clock) of the first and the last workgroup of ONE conv launch, from the HP_ABL=8
variant of the library: HIPPIE_HIP_LIB=tools/micro/variants/libhippie_abl8.so python tools/micro/conv_phases.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref, TapMap   # noqa: E402


def run(tm, flags, label):
    nb = tm.M // tm.Lout
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    a = put(nb * tm.Lin * tm.K * 4); w = put(3 * tm.N * tm.K * 4); out = put(tm.out_rows * tm.N * 4)
    bv = put(tm.N * 4); st = put(P.stat_repl(tm.N) * 2 * tm.N * 8); ts = put(512)
    buf = [a, w, out, bv, st] + [None] * 15 + [ts]
    ol = P.OpList()
    ol.add(P.CONV_TAPS, flags, tm.conv_ints(), (), buf)
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev[: w.offset + 3 * tm.N * tm.K * 4].view(torch.float32).normal_()
    rec = ol.array()[0]
    s = torch.cuda.current_stream().cuda_stream
    rows = []
    for _ in range(6):
        P.run_single_op(rec, [dev.data_ptr()] + [0] * 5, s)
        torch.cuda.synchronize()
        t = dev[ts.offset: ts.offset + 192].view(torch.int64).cpu().numpy().astype(np.int64)
        rows.append(t.copy())
    t = rows[-1]
    first, last = t[:5], t[8:13]
    ns = lambda x: (x - first[0]) * 10
    print(f"{label}: first WG  start 0, loop from {ns(first[1])}, loop end {ns(first[2])}, reduced {ns(first[3])}, stores issued {ns(first[4])} ns")
    cyc = t[16:21]
    nsteps = len(tm.taps) * tm.K // 32
    print(f"{' ' * len(label)}  loop: {cyc[2] - cyc[1]} shader cycles in {ns(first[2]) - ns(first[1])} ns = {(cyc[2] - cyc[1]) / max(1, ns(first[2]) - ns(first[1])):.2f} GHz, {(cyc[2] - cyc[1]) / nsteps:.0f} cycles per K-step (1024 = the MFMA pipe's own time)")
    print(f"{' ' * len(label)}  last WG   start {ns(last[0])}, loop from {ns(last[1])}, loop end {ns(last[2])}, reduced {ns(last[3])}, stores issued {ns(last[4])} ns", flush=True)


for name, (B, L, N, K) in {"L4 M=2048 N=512 K=512": (512, 4, 512, 512), "L1 M=12800 N=64 K=64": (512, 25, 64, 64)}.items():
    for nt, kk in ((1, 32), (3, K)):
        tm = TapMap(B * L, N, kk, L, L, L, 1, 0, [((t % 3) - 1, t % 3) for t in range(nt)])
        run(tm, 0, f"{name} {nt * kk // 32:2d} steps plain")
        run(tm, P.CONV_STATS, f"{name} {nt * kk // 32:2d} steps stats")
