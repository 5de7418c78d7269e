"""Debug: why does the zipped pair deviate from the single engines at iteration 2 when a graph-captured
forward/backward ran before the loop?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hippie_amd import planner
from hippie_amd.engine import Engine
from hippie_amd.pair import PairEngine
from oracle import cvae_oracle as O

def run(probe, probe_graph, restore):
    z, B = 10, 24
    cfgs = [planner.ModelCfg("unimodal", z, 50), planner.ModelCfg("unimodal", z, 100)]
    tcs = [planner.TrainCfg(lr=1e-6, clip=0.0), planner.TrainCfg(lr=1e-6, clip=1.0)]
    pe = PairEngine(cfgs[0], cfgs[1], B, tcs[0], tcs[1])
    singles = [Engine(c, B, t) for c, t in zip(cfgs, tcs)]
    for k, L in enumerate((50, 100)):
        om = O.OracleModel("unimodal", z, L, salt=20 + k)
        sd = {kk: v.detach() for kk, v in om.state.items()}
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=20 + k)
        for e in (pe.models[k], singles[k]):
            e.load_state_dict(sd)
            e.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
    if probe:
        pe.forward(True, probe_graph)
        pe.backward(probe_graph)
        torch.cuda.synchronize()
        if restore:
            for k, e in enumerate(pe.models):
                e.load_state_dict(singles[k].state_dict())
    print(f"--- probe={probe} probe_graph={probe_graph} restore={restore}")
    for it, use_graph in enumerate((False, True, True)):
        pe.train_step(use_graph=use_graph)
        for e in singles:
            e.train_step(use_graph=False)
        torch.cuda.synchronize()
        for k in range(2):
            a, b = pe.models[k], singles[k]
            print(f" it {it} m{k}: loss {a.scalars()[0]:.6f} / {b.scalars()[0]:.6f}  step {a.adam_step}/{b.adam_step} "
                  f"|p| {float(a.params.double().abs().sum()):.6f}/{float(b.params.double().abs().sum()):.6f} "
                  f"|g| {float(a.grads.double().abs().sum()):.6e}/{float(b.grads.double().abs().sum()):.6e} "
                  f"|m| {float(a.m.double().abs().sum()):.6e}/{float(b.m.double().abs().sum()):.6e} "
                  f"|v| {float(a.v.double().abs().sum()):.6e}/{float(b.v.double().abs().sum()):.6e}")

run(False, False, False)
run(True, True, True)
run(True, False, True)
run(True, True, False)
