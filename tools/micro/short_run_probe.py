"""Where does a SHORT timed region (the driver's --steps 20 --warmup 5) lose time against the steady state?  Per-step completion
times of both model streams (events), for the bench's own Pair.
python tools/micro/short_run_probe.py [steps] [warmup] [run-ahead]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # host run-ahead bound (0 = unbounded)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
data = bench.synth_dataset(bench.N_UNITS, dev, lw=50, lt=100)
pair = bench.Pair(dev, 1)
g = torch.Generator(device="cpu").manual_seed(1234)
perm = torch.randperm(bench.N_UNITS, generator=g).to(dev)
pair.load_tables(data, perm)
pair.pick_streams()
for rep in range(3):
    pair.fork()
    for i in range(W):
        pair.step(data, None, True)
    pair.join()
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(K)] for _ in pair.eng]
    t0e = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    t0e.record()
    pair.fork()
    host = []
    for i in range(K):
        if D and i >= D:
            for k in range(len(pair.streams)):
                ev[k][i - D].synchronize()
        pair.step(data, None, True)
        for k, s in enumerate(pair.streams):
            ev[k][i].record(s)
        host.append((time.perf_counter() - t0) * 1e3)
    pair.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    ends = [[t0e.elapsed_time(e) for e in evs] for evs in ev]
    print(f"run {rep}: {dt:.2f} ms for {K} steps = {dt / K:.3f} ms/step; host finished launching at {host[-1]:.2f} ms")
    for k, name in enumerate(("wave", "time")):
        d = [ends[k][0]] + [ends[k][i] - ends[k][i - 1] for i in range(1, K)]
        if K <= 40:
            print(f"  {name}: step durations (ms) " + " ".join(f"{v:.2f}" for v in d) + f" | last ends at {ends[k][-1]:.2f}")
        else:
            print(f"  {name}: mean step duration per block of 20 (ms) " + " ".join(f"{sum(d[i:i + 20]) / len(d[i:i + 20]):.3f}" for i in range(0, K, 20)))
    print("  host launch-done times (ms): " + " ".join(f"{v:.1f}" for v in host[:8]) + " ...")
