"""1-rank RCCL probe: what does one dist.all_reduce of the 32 MB gradient arena cost on this runtime,
by op (AVG / SUM), sync style and size?  (tools/, not part of the product.)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (8056640, 8056640 // 8):
    g = torch.randn(n, device="cuda")
    for name, fn in (("AVG", lambda: dist.all_reduce(g, op=dist.ReduceOp.AVG)),
                     ("SUM", lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM)),
                     ("SUM async+wait", lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True).wait())):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        t_host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        print(f"n={n:9d} {name:16s} device {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us/call   host enqueue {t_host * 1e6:8.1f} us/call", flush=True)
dist.destroy_process_group()
