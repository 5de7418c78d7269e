"""Does alternating between different kernels cost more per launch than repeating one (instruction-cache refills)?
Graphs of 200 conv launches of one shape: AAAA..., BBBB..., ABAB... (A = [N][K] weights, B = [K][N] weights: two
instantiations of the conv kernel), and a 4-kernel rotation with the IN_BN / stats variants."""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools/micro")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref, TapMap   # noqa: E402

REP = 200


def run(tm, flag_seq):
    nb = tm.M // tm.Lout
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    K, N = tm.K, tm.N
    a = put(nb * tm.Lin * K * 4); w = put(3 * N * K * 4); out = put(tm.M * N * 4); st_out = put(P.stat_repl(N) * 2 * N * 8)
    gamma, beta, rm, rv = put(K * 4), put(K * 4), put(K * 4), put(K * 4)
    st_in = put(P.stat_repl(K) * 2 * K * 8); save = put(2 * K * 4); coef = put(2 * K * 4)
    ol = P.OpList()
    for i in range(REP):
        fl = flag_seq[i % len(flag_seq)]
        if fl & P.CONV_IN_BN:
            ol.add(P.CONV_TAPS, fl | P.CONV_STATS, tm.conv_ints() + [nb * tm.Lin, 0], [0, 0, 0.01, 1e-5, 0.1],
                   [a, w, out, None, st_out, gamma, beta, rm, rv, None, None, None, st_in, save, coef])
        else:
            ol.add(P.CONV_TAPS, fl, tm.conv_ints(), (), [a, w, out, None, st_out])
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev[: st_out.offset].view(torch.float32).normal_()
    dev[gamma.offset: gamma.offset + 4 * K * 4].view(torch.float32).uniform_(0.5, 1.5)
    prog = P.DeviceProgram(ol.array(), [dev.data_ptr()] + [0] * 5, [dev.numel(), 4, 4, 4, 4, 4])
    seg = prog.capture(0, REP)
    s = torch.cuda.current_stream().cuda_stream
    prog.replay(seg, s); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.replay(seg, s); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    prog.close()
    return best


for name, (B, L, N, K) in {"L1 M=12800 N=64 K=64 (6 K-steps)": (512, 25, 64, 64), "L3 M=3584 N=256 K=256 (24 K-steps)": (512, 7, 256, 256)}.items():
    tm = TapMap(B * L, N, K, L, L, L, 1, 0, [(t - 1, t) for t in range(3)])
    A, Bk, S, I = 0, P.CONV_W_KN, P.CONV_STATS, P.CONV_IN_BN
    ta, tb, ts, ti = run(tm, [A]), run(tm, [Bk]), run(tm, [S]), run(tm, [I])
    tab = run(tm, [A, Bk])
    t4 = run(tm, [A, Bk, S | Bk, I])
    print(f"{name}: A {ta:.2f}  B {tb:.2f}  ABAB {tab:.2f} (mean of A,B {(ta + tb) / 2:.2f})   4-kernel rotation {t4:.2f} (mean of its members "
          f"{(ta + tb + run(tm, [S | Bk]) + ti) / 4:.2f})", flush=True)
