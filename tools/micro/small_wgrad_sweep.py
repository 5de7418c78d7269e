"""STEM_WGRAD / TAIL_BWD_W back to back in a captured graph; HIPPIE_WG_ROWS=<rows per thread> overrides the default."""
import os
import sys
import torch
sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref           # noqa: E402

REP = 100


def chain(build):
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    ol = P.OpList()
    build(ol, put)
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev.view(torch.float32)[: off // 4].normal_()
    prog = P.DeviceProgram(ol.array(), [dev.data_ptr()] + [0] * 5, [dev.numel(), 4, 4, 4, 4, 4])
    seg = prog.capture(0, REP)
    s = torch.cuda.current_stream().cuda_stream
    prog.replay(seg, s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.replay(seg, s); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    prog.close()
    return best


out = []
for L in (50, 100):
    B, L1 = 512, (L - 1) // 2 + 1
    def stem(ol, put):
        dr, x, dw = put(B * L1 * 64 * 4), put(B * L * 4), put(64 * 3 * 4)
        for _ in range(REP):
            ol.add(P.STEM_WGRAD, 0, [B, L, L1, 64], (), [dr, x, dw])
    out.append(f"stem L={L} {chain(stem):.2f}")
def tail(ol, put):
    B = 512
    dt, act, dw, db = put(B * 64 * 4), put(B * 32 * 64 * 4), put(64 * 3 * 4), put(4)
    for _ in range(REP):
        ol.add(P.TAIL_BWD_W, 0, [B, 32, 64], (), [dt, act, dw, db])
out.append(f"tail {chain(tail):.2f}")
print("rows per thread:", os.environ.get("HIPPIE_WG_ROWS", "default"), " ".join(out), flush=True)
