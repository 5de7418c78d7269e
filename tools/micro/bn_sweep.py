"""BatchNorm passes (apply / backward reduce / backward apply) back to back in a captured graph, per launch:
python tools/micro/bn_sweep.py   (HIPPIE_HIP_LIB = a variant built with -DHP_BN_ROWS=n)"""
import os
import sys
import torch
sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref           # noqa: E402

REP = 100


def chain(build, reps=REP):
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    ol = P.OpList()
    build(ol, put, reps)
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev.view(torch.float32)[: off // 4].normal_()
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


def main():
    print(os.environ.get("HIPPIE_HIP_LIB", "product library"), flush=True)
    for M, C in ((16384, 64), (2048, 512), (3584, 512), (512, 20), (512, 10)):
        R = P.stat_repl(C)
        def apply(ol, put, reps, res, training=1):
            raw, out, st = put(M * C * 4), put(M * C * 4), put(R * 2 * C * 8)
            g, b, rm, rv, save, rs = put(C * 4), put(C * 4), put(C * 4), put(C * 4), put(2 * C * 4), put(M * C * 4)
            for _ in range(reps):
                ol.add(P.BN_APPLY, 0, [M, C, 1 if res else 0, training, 1], [0.01, 1e-5, 0.1], [raw, out, st, g, b, rm, rv, save] + ([rs] if res else []))
        def bapply(ol, put, reps):
            gout, raw, save, bs = put(M * C * 4), put(M * C * 4), put(2 * C * 4), put(R * 2 * C * 8)
            g, dr, dg, db = put(C * 4), put(M * C * 4), put(C * 4), put(C * 4)
            for _ in range(reps):
                ol.add(P.BN_BWD_APPLY, 0, [M, C], (), [gout, raw, save, bs, g, dr, dg, db])
        def breduce(ol, put, reps):
            g1, out, gout, raw, save, bs = put(M * C * 4), put(M * C * 4), put(M * C * 4), put(M * C * 4), put(2 * C * 4), put(R * 2 * C * 8)
            for _ in range(reps):
                ol.add(P.BN_BWD_REDUCE, 0, [M, C, 0, 0], [0.01], [g1, None, out, gout, raw, save, bs])
        mb = M * C * 4 / 1e6
        t1 = chain(lambda ol, put, reps: apply(ol, put, reps, False))
        t2 = chain(lambda ol, put, reps: apply(ol, put, reps, True))
        t0 = chain(lambda ol, put, reps: apply(ol, put, reps, False, 0))
        t3 = chain(bapply)
        t4 = chain(breduce)
        print(f"  M={M:5d} C={C:3d} ({mb:5.2f} MB/tensor): eval-apply {t0:5.2f} us  apply {t1:5.2f} us ({2 * mb / t1:5.2f} TB/s)  apply+res {t2:5.2f} us ({3 * mb / t2:5.2f} TB/s)  "
              f"bwd-apply {t3:5.2f} us ({3 * mb / t3:5.2f} TB/s)  bwd-reduce {t4:5.2f} us ({4 * mb / t4:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
