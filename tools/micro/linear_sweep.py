"""The heads' weight-gradient launches (HP_OP_LINEAR_BWD_W) back to back in a captured graph, per launch, for the shapes
of the batch-512 unimodal model; HIPPIE_LBW_ROWS=<rows per M-slice> overrides the slicing heuristic."""
import os
import sys
import torch
sys.path.insert(0, ".")
from hippie_amd import program as P          # noqa: E402
from hippie_amd.program import Ref           # noqa: E402

REP = 100
SHAPES = {"linear_out 100x64 (M=512)": (512, 100, 64), "decoder.linear 512x20": (512, 512, 20), "decoder_fc 20x20": (512, 20, 20),
          "z_mean|logvar 20x10": (512, 20, 10), "encoder_fc.3 10x20": (512, 10, 20), "encoder_fc.0 20x30": (512, 20, 30),
          "encoder.linear 20x512": (512, 20, 512)}


def chain(M, N, K):
    off = 0
    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    dy, x, dw, db = put(M * N * 4), put(M * K * 4), put(N * K * 4), put(N * 4)
    ol = P.OpList()
    for _ in range(REP):
        ol.add(P.LINEAR_BWD_W, 0, [M, N, K, N, K], (), [dy, x, dw, db])
    dev = torch.zeros(off + 256, dtype=torch.uint8, device="cuda")
    dev.view(torch.float32)[: dw.offset // 4].normal_()
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


print("rows per slice:", os.environ.get("HIPPIE_LBW_ROWS", "heuristic"), " ".join(f"{n.split()[0]} {chain(*s):.2f}" for n, s in SHAPES.items()), flush=True)
