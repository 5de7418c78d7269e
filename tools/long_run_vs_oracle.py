"""Long-run sanity: the engine and the CPU oracle (torch fp32 restatement, pinned by the reference fixtures) train the SAME model on the SAME
batches and noise for many steps.  Individual trajectories diverge chaotically after a few dozen steps (leaky-ReLU branches, Adam), so
the check is on the loss CURVES: both must fall the same way (reconstruction error and total loss at every checkpoint within a factor,
the final plateau within [0.6, 1.67]).  Guards against errors that single-step parity cannot see (bias correction at large step counts, weight decay, the
KL weight, running statistics).        python tools/long_run_vs_oracle.py [steps] [batch] [L] [clip | 0] [f32 | bf16] [multimodal]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hippie_amd import planner
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O



def run(steps=240, B=128, z=10, L=50, lr=1e-3, pool=2048, verbose=True, clip=None, mfma_dtype=None, multimodal=False, act_dtype="f32"):
    """-> [(step, engine (loss, mse1, mse2, kl), oracle (loss, mse, kl))] at ~12 checkpoints"""
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    L2 = 100
    wave, isi, labels = bench.synth_dataset(pool, "cpu", lw=L, lt=L2 if multimodal else L)
    if clip and not multimodal:    # the time model of the pipeline: spike-timing histograms, gradient clipping (scripts/...:222)
        wave = isi
    kind = "multimodal" if multimodal else "unimodal"
    om = O.OracleModel(kind, z, L, output_size2=L2 if multimodal else None, salt=1)
    eng = Engine(planner.ModelCfg(kind, z, L, L2), B, planner.TrainCfg(lr=lr, weight_decay=0.01, beta=1.0, clip=clip or 0.0, act_dtype=act_dtype, **({} if mfma_dtype is None else dict(mfma_dtype=mfma_dtype))))
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    g = torch.Generator().manual_seed(5)
    rows = []
    t0 = time.time()
    for i in range(steps):
        idx = torch.randperm(pool, generator=g)[:B]
        x, src = wave[idx].view(B, 1, L), labels[idx]
        eps = torch.randn(B, z, generator=g)
        x2 = isi[idx].view(B, 1, L2) if multimodal else None
        eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda(), x2=x2.cuda() if multimodal else None)
        eng.train_step(True)
        outs, ls, _ = om.train_step((x, x2, src, None) if multimodal else (x, src, None), eps, lr, weight_decay=0.01, beta=1.0, clip=clip)
        if i % max(1, steps // 12) == 0 or i == steps - 1:
            e = eng.scalars()
            o = [float(v.detach()) for v in ls]
            rows.append((i, e, o))
            if verbose:
                print(f"step {i:4d}  engine loss {e[0]:.5f} mse {e[1]:.5f} kl {e[3]:.5f}   oracle loss {o[0]:.5f} mse {o[1]:.5f} kl {o[-1]:.5f}   ({time.time() - t0:.0f} s)", flush=True)
    return rows


def check(rows):
    """reconstruction error within a factor of 2.5 and loss within a factor of 3 of the oracle's at every checkpoint (one batch's KL term
    fluctuates by +-50 % once it is small); the plateau (mean of the last four checkpoints) within [0.6, 1.67]"""
    bad = []
    for i, e, o in rows:
        for name, a, b, f in (("loss", e[0], o[0], 3.0), ("mse", e[1], o[1], 2.5)):
            if not (b / f - 1e-5 <= a <= f * b + 1e-5):
                bad.append((i, name, a, b))
    tail_e = sum(r[1][0] for r in rows[-4:]) / 4
    tail_o = sum(r[2][0] for r in rows[-4:]) / 4
    return bad, tail_e, tail_o


if __name__ == "__main__":
    rows = run(int(sys.argv[1]) if len(sys.argv) > 1 else 240, int(sys.argv[2]) if len(sys.argv) > 2 else 128,
               L=int(sys.argv[3]) if len(sys.argv) > 3 else 50, clip=(float(sys.argv[4]) or None) if len(sys.argv) > 4 else None,
               mfma_dtype=sys.argv[5] if len(sys.argv) > 5 else None, multimodal=len(sys.argv) > 6 and sys.argv[6] == "multimodal")
    bad, tail_e, tail_o = check(rows)
    print(f"plateau (last four checkpoints): engine {tail_e:.5f}, oracle {tail_o:.5f}, ratio {tail_e / tail_o:.3f}")
    assert not bad, bad
    assert 0.6 <= tail_e / tail_o <= 1.67
    print("long run ok")
