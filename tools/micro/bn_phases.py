"""Phase timestamps of one BN_APPLY launch (library built with -DHP_BN_TS)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref           # noqa: E402

for M, C in ((16384, 64), (2048, 512)):
    off = 0
    def put(nbytes):
        global off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    R = P.stat_repl(C)
    raw, out, st = put(M * C * 4), put(M * C * 4), put(R * 2 * C * 8)
    g, b, rm, rv, save = put(C * 4), put(C * 4), put(C * 4), put(C * 4), put(2 * C * 4)
    ts = put(256)
    ol = P.OpList()
    ol.add(P.BN_APPLY, 0, [M, C, 0, 1, 1], [0.01, 1e-5, 0.1], [raw, out, st, g, b, rm, rv, save, None, None, None, None, None, None, ts])
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev.view(torch.float32)[: save.offset // 4].normal_()
    rec = ol.array()[0]
    for _ in range(5):
        P.run_single_op(rec, [dev.data_ptr()] + [0] * 5, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    t = dev[ts.offset: ts.offset + 128].view(torch.int64).cpu().numpy()
    f, l = t[:4], t[8:12]
    ns = lambda x: int(x - f[0]) * 10
    print(f"M={M} C={C}: first block: coef done {ns(f[1])}, past barrier {ns(f[2])}, stores issued {ns(f[3])} ns | last block: start {ns(l[0])}, coef {ns(l[1])}, barrier {ns(l[2])}, stores {ns(l[3])} ns", flush=True)
