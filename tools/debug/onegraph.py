import sys, time, torch
sys.path.insert(0, ".")
from hippie_amd import planner
from hippie_amd.engine import Engine
for L, clip in ((100, 1.0), (50, 0.0)):
    e = Engine(planner.ModelCfg(kind="unimodal", z_dim=10, output_size=L), 512, planner.TrainCfg(lr=1e-3, clip=clip))
    g = torch.Generator().manual_seed(0)
    e.set_inputs(torch.randn(512, 1, L, generator=g).cuda(), torch.randint(0, 5, (512,), generator=g).cuda())
    segs = e.plan.ops.segments
    f0, fc = segs["fwd_train"]; b0, bc = segs["bwd"]; o0, oc = segs["opt"]
    assert f0 + fc == b0 and b0 + bc == o0
    segs["step"] = (f0, fc + bc + oc)
    for mode in ("three graphs", "one graph"):
        def step():
            if mode == "one graph":
                e.run("step", True)
            else:
                e.run("fwd_train", True); e.run("bwd", True); e.run("opt", True)
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): step()
        torch.cuda.synchronize(); print(f"L={L} {mode}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms/step")
