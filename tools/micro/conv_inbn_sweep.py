"""Plain conv vs HP_CONV_IN_BN conv (BatchNorm + leaky_relu of the input in the operand loader) in a captured graph:
fixed cost (coefficient derivation in every workgroup) and cost per K-step of the loader transform."""
import sys
import torch
sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref, TapMap   # noqa: E402

REP = 100


def time_op(tm, in_bn, reps=REP):
    nb = tm.M // tm.Lout
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    K, N = tm.K, tm.N
    a = put(nb * tm.Lin * K * 4); w = put(3 * N * K * 4); out = put(tm.M * N * 4)
    st_out = put(P.stat_repl(N) * 2 * N * 8)
    gamma, beta, rm, rv = put(K * 4), put(K * 4), put(K * 4), put(K * 4)
    st_in = put(P.stat_repl(K) * 2 * K * 8); save = put(2 * K * 4); coef = put(2 * K * 4)
    ol = P.OpList()
    for _ in range(reps):
        if in_bn:
            ol.add(P.CONV_TAPS, P.CONV_STATS | P.CONV_IN_BN, tm.conv_ints() + [nb * tm.Lin, 0], [0, 0, 0.01, 1e-5, 0.1],
                   [a, w, out, None, st_out, gamma, beta, rm, rv, None, None, None, st_in, save, coef])
        else:
            ol.add(P.CONV_TAPS, P.CONV_STATS, tm.conv_ints(), (), [a, w, out, None, st_out])
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev[: st_out.offset].view(torch.float32).normal_()
    dev[gamma.offset: gamma.offset + 4 * K * 4].view(torch.float32).uniform_(0.5, 1.5)
    prog = P.DeviceProgram(ol.array(), [dev.data_ptr()] + [0] * 5, [dev.numel(), 4, 4, 4, 4, 4])
    seg = prog.capture(0, reps)
    s = torch.cuda.current_stream().cuda_stream
    prog.replay(seg, s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.replay(seg, s); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    prog.close()
    return best


for name, (B, L, N, K) in {"L4 M=2048 N=512 K=512": (512, 4, 512, 512), "L4t M=3584 N=512 K=512": (512, 7, 512, 512),
                           "L2 M=6656 N=128 K=128": (512, 13, 128, 128), "L1 M=12800 N=64 K=64": (512, 25, 64, 64)}.items():
    for in_bn in (False, True):
        row = []
        for nt in (1, 3, 6):
            tm = TapMap(B * L, N, K, L, L, L, 1, 0, [((t % 3) - 1, t % 3) for t in range(nt)])
            row.append((nt * K // 32, time_op(tm, in_bn)))
        (s0, t0), (s1, t1) = row[0], row[-1]
        slope = (t1 - t0) / (s1 - s0)
        print(f"{name} {'in_bn' if in_bn else 'plain'} " + " ".join(f"{s:3d} steps {t:6.2f} us" for s, t in row) +
              f" | {slope * 1e3:6.1f} ns/step, fixed {t0 - slope * s0:5.2f} us", flush=True)
