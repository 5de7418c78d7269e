"""Does the depth of the host's run-ahead change the GPU's step time?  short_run_probe.py: while the host is still enqueuing (a 400-step
run fills the queues for the first ~300 steps) a pair-step takes 4.31-4.35 ms; once everything is queued, 4.45 ms — and a 20-step run is
queued within 3 ms.  Here: K steps with the host allowed to be at most D steps ahead of the GPU (waits on the event of step i - D).
python tools/micro/runahead_probe.py [steps]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
data = bench.synth_dataset(bench.N_UNITS, dev, lw=50, lt=100)
pair = bench.Pair(dev, 1)
g = torch.Generator(device="cpu").manual_seed(1234)
perm = torch.randperm(bench.N_UNITS, generator=g).to(dev)
pair.load_tables(data, perm)
pair.pick_streams()


def run(D, spin):
    pair.fork()
    for i in range(5):
        pair.step(data, None, True)
    pair.join()
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event() for _ in range(K)] for _ in pair.eng]
    t0 = time.perf_counter()
    pair.fork()
    for i in range(K):
        if D and i >= D:
            for k in range(len(pair.eng)):
                if spin:
                    while not ev[k][i - D].query():
                        pass
                else:
                    ev[k][i - D].synchronize()
        pair.step(data, None, True)
        if D:
            for k, s in enumerate(pair.streams):
                ev[k][i].record(s)
    pair.join()
    if spin:
        fin = torch.cuda.Event()
        fin.record()
        while not fin.query():
            pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


for D, spin in ((0, False), (0, True), (1, False), (1, True), (2, False), (2, True), (3, True), (4, True), (8, True), (0, False)):
    run(D, spin)
    v = min(run(D, spin) for _ in range(3))
    print(f"K={K} run-ahead {'unbounded' if D == 0 else D} {'spin ' if spin else 'block'}: {v:.3f} ms per pair-step", flush=True)
