"""GPU: the whole engine (planner -> libhippie_hip.so -> MI355X) against the CPU oracle and the
golden vectors generated from the reference's own modules."""
import dataclasses
import os
import re

import numpy as np
import pytest
import torch

from hippie_amd import planner, program as P
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(8)


def n(t):
    return t.detach().cpu().numpy()


def build(kind, z, L, B, with_class, beta, clip, lr, salt, L2=None, w1=1.0, w2=1.0, fuse_bn=True):
    cfg = planner.ModelCfg(kind=kind, z_dim=z, output_size=L, output_size2=L2 or 100)
    tc = planner.TrainCfg(lr=lr, weight_decay=0.01, beta=beta, clip=clip or 0.0, w1=w1, w2=w2,
                          fuse_bn=fuse_bn)
    eng = Engine(cfg, B, tc, with_class=with_class)
    oms = []
    for dt in (torch.float32, torch.float64):
        oms.append(O.OracleModel(kind, z, L, output_size2=L2, salt=salt, dtype=dt))
    eng.load_state_dict({k: v.detach() for k, v in oms[0].state.items()})
    if kind == "unimodal":
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt)
        batch = (x, src, cls if with_class else None)
        eng.set_inputs(x.cuda(), src.cuda(), cls.cuda() if with_class else None, eps.cuda())
    else:
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt, name="x1")
        x2, _, _, _ = O.synth_inputs(B, L2, z, salt=salt, name="x2")
        batch = (x, x2, src, cls if with_class else None)
        eng.set_inputs(x.cuda(), src.cuda(), cls.cuda() if with_class else None, eps.cuda(), x2=x2.cuda())
    batch64 = tuple(t.double() if (t is not None and t.is_floating_point()) else t for t in batch)
    return eng, oms, batch, batch64, eps


def check_forward(eng, oms, batch, batch64, eps, training):
    with torch.no_grad():
        o32 = oms[0].forward(batch, eps, training=training)
        o64 = oms[1].forward(batch64, eps.double(), training=training)
    outs = eng.forward(training=training)
    torch.cuda.synchronize()
    names = ["enc", "mu", "logvar", "rec", "rec2"]
    for k, (a, b, c) in enumerate(zip(outs, o32, o64)):
        H.parity(n(a).reshape(n(b).shape), n(b), n(c), f"{'train' if training else 'eval'} {names[k]}")
    return o32, o64


CASES = {
    "wave": dict(kind="unimodal", z=10, L=50, B=16, with_class=False, beta=1.0, clip=None, lr=1e-3, salt=0),
    "time_clip": dict(kind="unimodal", z=10, L=100, B=16, with_class=False, beta=1.0, clip=1.0, lr=1e-3, salt=0),
    "cls_z5": dict(kind="unimodal", z=5, L=50, B=12, with_class=True, beta=0.5, clip=1.0, lr=1e-4, salt=3),
    "z32_L256": dict(kind="unimodal", z=32, L=256, B=8, with_class=False, beta=1.0, clip=None, lr=1e-3, salt=5),
    "z32_L32": dict(kind="unimodal", z=32, L=32, B=8, with_class=False, beta=1.0, clip=None, lr=1e-3, salt=6),
    "multi": dict(kind="multimodal", z=10, L=50, L2=100, B=12, with_class=False, beta=1.0, clip=1.0, lr=1e-3, salt=7, w1=1.0, w2=0.5),
    # BASELINE config 5's per-rank model (multimodal, z=64, wave 256 + time 32) at the fixture's tiny batch
    "multi_c5": dict(kind="multimodal", z=64, L=256, L2=32, B=8, with_class=False, beta=1.0, clip=1.0, lr=1e-3, salt=8, w1=1.0, w2=1.0),
    # the optional lowering: one launch per BatchNorm pass (no loader / epilogue fusion)
    "cls_z5_unfused_bn": dict(kind="unimodal", z=5, L=50, B=12, with_class=True, beta=0.5, clip=1.0, lr=1e-4, salt=3, fuse_bn=False),
}
GOLDEN = {"wave": "unimodal_wave_z10_L50_B16.npz", "time_clip": "unimodal_time_z10_L100_B16_clip.npz",
          "cls_z5": "unimodal_wave_z5_L50_B12_cls.npz", "z32_L256": "unimodal_wave_z32_L256_B8.npz",
          "z32_L32": "unimodal_time_z32_L32_B8.npz", "multi": "multimodal_z10_B12.npz", "multi_c5": "multimodal_z64_L256_32_B8.npz",
          "cls_z5_unfused_bn": "unimodal_wave_z5_L50_B12_cls.npz"}


def masked_oracle_step(eng, oms, batch, batch64, eps, c, tag):
    """Engine training forward + backward, then BOTH oracles (float32 and float64) re-evaluated on the leaky-ReLU
    branches the engine actually took (helpers.engine_masks -> OracleModel.forward(masks=...)): the comparison of
    gradients is then decidable at the 1e-4 bar whatever the number of near-zero activations.  Before that, the engine's
    branches themselves are held against the free-running oracle (flip budget).  Returns
    (engine outputs, oracle32 outs/losses, oracle64 outs/losses, flips vs the UNMASKED float64 oracle)."""
    for om in oms:
        for k in om.param_keys:
            om.state[k].grad = None
    outs = eng.forward(True)
    eng.backward()
    torch.cuda.synchronize()
    pres = H.engine_pre_activations(eng)
    masks = {k: torch.from_numpy(v > 0) for k, v in pres.items()}
    # The UNMASKED anchor: against the FREE-RUNNING oracles (throw-away copies: a training forward mutates the running
    # statistics) every leaky-ReLU input tensor of the engine meets the parity criterion, at most 3e-6 of the signs differ,
    # and every element whose sign differs is closer to zero than float32 resolves (helpers.assert_flip_budget).
    # Only then are both oracles re-evaluated on the engine's branches for the gradient comparison.
    taps_free = []
    for dt, b_, e_ in ((torch.float32, batch, eps), (torch.float64, batch64, eps.double())):
        free = O.OracleModel(c["kind"], c["z"], c["L"], output_size2=c.get("L2"), salt=c["salt"], dtype=dt)
        taps_free.append({})
        with torch.no_grad():
            free.forward(b_, e_, True, taps=taps_free[-1])
    flips, elems, worst = H.assert_flip_budget(pres, taps_free[0], taps_free[1], tag)
    print(f"[{tag}] leaky-ReLU inputs whose sign differs from the free-running f64 oracle: {flips} of {elems} "
          f"(budget {H.flip_allowance(elems)}); largest |pre| among them {worst:.2e} of its tensor's max")
    outs32 = oms[0].forward(batch, eps, True, masks=masks)
    ls32 = oms[0].losses(batch, outs32, c["beta"], c.get("w1", 1.0), c.get("w2", 1.0))
    ls32[0].backward()
    outs64 = oms[1].forward(batch64, eps.double(), True, masks=masks)
    ls64 = oms[1].losses(batch64, outs64, c["beta"], c.get("w1", 1.0), c.get("w2", 1.0))
    ls64[0].backward()
    return outs, (outs32, ls32), (outs64, ls64), flips


def check_all_gradients(eng, oms, tag):
    """EVERY gradient tensor within 1e-4 (relative to the tensor's max) of the masked float64 oracle, or within
    3x the masked float32 oracle's own distance to it (helpers.parity).  No loose fallback."""
    grads = eng.grad_dict()
    g32, g64 = oms[0].grads(), oms[1].grads()
    worst = 0.0
    for k, gr in g32.items():
        mine = n(grads[k])
        if gr is None:
            assert np.all(mine == 0), k
            continue
        if re.search(H.ZERO_GRAD_RE, k):   # analytically zero: both sides hold rounding noise only
            continue
        e, _ = H.parity(mine, n(gr), n(g64[k]), f"{tag} grad {k}")
        worst = max(worst, e)
    print(f"[{tag}] worst gradient error vs masked f64 oracle: {worst:.3e}")
    return worst


@pytest.mark.parametrize("name", list(CASES))
def test_forward_grads_and_step_vs_oracle(name):
    c = CASES[name]
    eng, oms, batch, batch64, eps = build(**c)
    check_forward(eng, oms, batch, batch64, eps, training=False)
    outs, (outs32, ls32), (outs64, ls64), flips = masked_oracle_step(eng, oms, batch, batch64, eps, c, name)
    names = ["enc", "mu", "logvar", "rec", "rec2"]
    for k, (a, b, cc) in enumerate(zip(outs, outs32, outs64)):
        H.parity(n(a).reshape(n(b).shape), n(b), n(cc), f"train {names[k]}")
    sc = eng.scalars()
    want = [float(v) for v in ls64]
    got = [sc[0], sc[1], sc[3]] if c["kind"] == "unimodal" else sc
    np.testing.assert_allclose(got, want, rtol=1e-4)
    # golden scalars straight from the reference's training_step
    g = dict(np.load(os.path.join(G, GOLDEN[name])))
    np.testing.assert_allclose(got, g["scalars"], rtol=1e-4)
    check_all_gradients(eng, oms, name)
    # running statistics
    sd = eng.state_dict()
    for k in eng.plan.bufs:
        H.parity(n(sd[k]), n(oms[0].state[k]), n(oms[1].state[k]), "buffer " + k, rel=1e-5)
        assert int(sd[k.rsplit(".", 1)[0] + ".num_batches_tracked"]) == 1
    # optimiser step on the engine's own gradients vs torch AdamW semantics on the (masked) oracle's
    eng.optimizer_step()
    torch.cuda.synchronize()
    assert eng.adam_step == 1
    with torch.no_grad():
        gd = oms[0].grads()
        if c["clip"]:
            O.clip_grad_norm(list(gd.values()), c["clip"])
        for k in oms[0].param_keys:
            if gd[k] is not None:
                oms[0].exp_avg[k] = torch.zeros_like(oms[0].state[k])
                oms[0].exp_avg_sq[k] = torch.zeros_like(oms[0].state[k])
        O.adamw_step({k: oms[0].state[k] for k in oms[0].param_keys}, gd, oms[0].exp_avg, oms[0].exp_avg_sq, 1, c["lr"], 0.01)
    sd = eng.state_dict()
    for k in oms[0].param_keys:
        if re.search(H.ZERO_GRAD_RE, k) or gd[k] is None:
            # noise-level gradients: Adam's first step is -lr*sign(g), so only the 2*lr bound is meaningful; the
            # kernel itself is checked against torch.optim.AdamW on identical inputs in test_gpu_ops.py
            assert np.abs(n(sd[k]) - n(oms[0].state[k])).max() <= 2.2 * c["lr"] + 1e-7, k
            continue
        H.assert_adam_close(n(sd[k]), n(oms[0].state[k]), c["lr"], k, grad=n(gd[k]))


TRAJ = {
    "wave": (dict(kind="unimodal", z=10, L=50, B=32, with_class=False, beta=1.0, clip=None, lr=1e-6, salt=9), "unimodal_wave_z10_L50_B32_traj.npz"),
    "time_clip": (dict(kind="unimodal", z=10, L=100, B=32, with_class=False, beta=1.0, clip=1.0, lr=1e-6, salt=10), "unimodal_time_z10_L100_B32_traj_clip.npz"),
}


@pytest.mark.parametrize("name", list(TRAJ))
@pytest.mark.parametrize("use_graph", [False, True])
def test_training_trajectory_vs_reference_golden(name, use_graph):
    """Six optimisation steps against the losses logged by the reference's own LightningModule +
    torch.optim.AdamW (+ clip_grad_norm_ for the time model), eager and through hipGraph replay."""
    c, fname = TRAJ[name]
    eng, oms, batch, batch64, eps = build(**c)
    g = dict(np.load(os.path.join(G, fname)))
    got = []
    for s_ in range(6):
        eng.train_step(use_graph=use_graph)
        sc = eng.scalars()
        got.append([sc[0], sc[1], sc[3]])
    got = np.array(got)
    dev = np.abs(got - g["scalars_traj"]) / np.abs(g["scalars_traj"])
    print(f"[traj {name} graph={use_graph}] max rel deviation per step: " + " ".join(f"{v:.2e}" for v in dev.max(1)))
    np.testing.assert_allclose(got[0], g["scalars_traj"][0], rtol=1e-4)
    # Later steps: UNMASKED, against what the reference itself logged.  Measured (MI355X, round 3): <= 9.8e-4 over the six
    # steps — at batch 32 at most one leaky-ReLU sign differs from the oracle's (helpers.assert_flip_budget), and what is left is
    # Adam's +-lr move of elements whose gradient is rounding noise in ANY float32 implementation, the reference's included
    # (test_masked_trajectory_is_tight holds the same steps to 5e-5 against the float64 oracle).  Bound: 1.5e-3 (round 2: 2e-3).
    np.testing.assert_allclose(got, g["scalars_traj"], rtol=1.5e-3)
    # the optimiser's effect: cumulative loss decrements agree with the reference's to 1 % (measured <= 0.22 %; round 2: 5 %)
    d_mine, d_ref = got[0, 0] - got[1:, 0], g["scalars_traj"][0, 0] - g["scalars_traj"][1:, 0]
    print(f"[traj {name} graph={use_graph}] loss decrements vs reference: max rel deviation {np.abs(d_mine / d_ref - 1).max():.2e}")
    np.testing.assert_allclose(d_mine, d_ref, rtol=1e-2)
    assert eng.adam_step == 6
    sd = eng.state_dict()
    worst_frac = 0.0
    for k in g:
        if k.startswith("param_step6.") and not re.search(H.ZERO_GRAD_RE, k):
            # fraction of a tensor's elements further from the reference's than Adam noise explains: measured <= 7.2e-3; bound 2e-2 (round 2: 5e-2)
            worst_frac = max(worst_frac, H.assert_adam_close(n(sd[k.split(".", 1)[1]]), g[k], c["lr"], k, steps=6, frac=2e-2))
    print(f"[traj {name} graph={use_graph}] parameters after 6 steps: worst fraction of elements beyond noise {worst_frac:.2e}")
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 6


FULL = {
    "wave": dict(kind="unimodal", z=10, L=50, B=512, with_class=False, beta=1.0, clip=None, lr=1e-3, salt=11),
    "time_clip": dict(kind="unimodal", z=10, L=100, B=512, with_class=False, beta=1.0, clip=1.0, lr=1e-3, salt=12),
    "multi": dict(kind="multimodal", z=10, L=50, L2=100, B=512, with_class=False, beta=1.0, clip=1.0, lr=1e-3, salt=13, w1=1.0, w2=0.5),
}


@pytest.mark.parametrize("name", list(FULL))
def test_full_batch_512_step_matches_oracle(name):
    """BASELINE config shape: batch 512, z=10 — forward, loss and EVERY gradient at full size, against the float64
    oracle evaluated on the engine's own leaky-ReLU branches (1e-4, no loose bound)."""
    c = FULL[name]
    eng, oms, batch, batch64, eps = build(**c)
    outs, (outs32, ls32), (outs64, ls64), flips = masked_oracle_step(eng, oms, batch, batch64, eps, c, "B512 " + name)
    for k, (a, b, cc) in enumerate(zip(outs, outs32, outs64)):
        H.parity(n(a).reshape(n(b).shape), n(b), n(cc), f"B512 out{k}")
    sc = eng.scalars()
    got = [sc[0], sc[1], sc[3]] if c["kind"] == "unimodal" else sc
    np.testing.assert_allclose(got, [float(v) for v in ls64], rtol=1e-4)
    check_all_gradients(eng, oms, "B512 " + name)


@pytest.mark.parametrize("name", ["wave", "time_clip"])
def test_masked_trajectory_is_tight(name):
    """Six optimisation steps, engine vs the float64 oracle stepping in lock step on the engine's own leaky-ReLU
    branches (re-read every step): losses to 1e-5 and, at the end, every parameter with a significant gradient
    to 1e-4 relative — the trajectory check without the mask-flip allowance of the golden-log test below.
    (The loss of this model falls by ~10 % per step even at lr 1e-5 — Adam moves all 8 M parameters by lr each —
    so a 1e-6 float32 difference in one step's loss grows to ~1e-5 within four steps; the bound is 5e-5, 40x tighter
    than what the unmasked comparison with the reference's logged losses can hold.)"""
    # lr 1e-5: Adam's update lr*m/(sqrt(v)+eps) is +-lr for EVERY element, including those whose gradient is at
    # float32 rounding level (their sign is arbitrary in any float32 implementation, the reference's included); at
    # lr 1e-3 those elements alone move the loss by 4e-5 per step, at 1e-5 by 4e-7
    c = dict(TRAJ[name][0], lr=1e-5)
    eng, oms, batch, batch64, eps = build(**c)
    om = oms[1]
    lr, clip = c["lr"], c["clip"]
    for step in range(6):
        for k in om.param_keys:
            om.state[k].grad = None
        eng.forward(True)
        eng.backward()
        torch.cuda.synchronize()
        masks = H.engine_masks(eng)
        outs64 = om.forward(batch64, eps.double(), True, masks=masks)
        ls64 = om.losses(batch64, outs64, c["beta"])
        ls64[0].backward()
        sc = eng.scalars()
        want = np.array([float(v.detach()) for v in ls64])
        print(f"[masked traj {name}] step {step}: max rel deviation {np.abs(np.array([sc[0], sc[1], sc[3]]) / want - 1).max():.2e}")
        np.testing.assert_allclose([sc[0], sc[1], sc[3]], want, rtol=5e-5, err_msg=f"step {step}")
        with torch.no_grad():
            g = om.grads()
            if clip:
                O.clip_grad_norm(list(g.values()), clip)
            for k in om.param_keys:
                if g[k] is not None and k not in om.exp_avg:
                    om.exp_avg[k] = torch.zeros_like(om.state[k])
                    om.exp_avg_sq[k] = torch.zeros_like(om.state[k])
            om.step_count += 1
            O.adamw_step({k: om.state[k] for k in om.param_keys}, g, om.exp_avg, om.exp_avg_sq, om.step_count, lr, 0.01)
        eng.optimizer_step()
    torch.cuda.synchronize()
    sd = eng.state_dict()
    g = om.grads()
    for k in om.param_keys:
        if g[k] is None:
            continue
        if re.search(H.ZERO_GRAD_RE, k):
            assert np.abs(n(sd[k]) - n(om.state[k])).max() <= 6 * 2.2 * lr, k
            continue
        # elements whose gradient stayed significant move identically; the few whose gradient crossed zero during the
        # six steps (Adam's +-lr regime) are bounded by assert_adam_close's 2.2*lr*steps and may number <= 0.1 %
        a, d_ = n(sd[k]).astype(np.float64).reshape(-1), n(om.state[k]).astype(np.float64).reshape(-1)
        assert np.abs(a - d_).max() <= 2.2 * lr * 6 + 1e-7, k
        gk = np.abs(n(g[k]).astype(np.float64).reshape(-1))
        bad = (np.abs(a - d_) > 1e-4 * np.abs(d_) + 0.02 * lr) & (gk > 1e-2 * gk.max())
        assert bad.mean() <= 1e-3, f"{k}: {bad.sum()} of {bad.size} elements with significant gradient differ"
    for k in eng.plan.bufs:
        H.assert_close(n(sd[k]), n(om.state[k]), 1e-4, k)        # running statistics follow the +-lr parameter noise


def test_graph_replay_equals_eager_and_is_repeatable():
    c = dict(CASES["wave"], lr=1e-5)
    eng, oms, batch, batch64, eps = build(**c)
    eng2 = Engine(eng.cfg, eng.B, eng.train_cfg, with_class=False)
    eng2.load_state_dict(eng.state_dict())
    x, src, cls, e = O.synth_inputs(c["B"], c["L"], c["z"], salt=c["salt"])
    eng2.set_inputs(x.cuda(), src.cuda(), None, e.cuda())
    for _ in range(3):
        eng.train_step(use_graph=False)
        eng2.train_step(use_graph=True)
    torch.cuda.synchronize()
    a, b = eng.state_dict(), eng2.state_dict()
    for k in a:
        if a[k].dtype.is_floating_point:
            if re.search(H.ZERO_GRAD_RE, k):
                assert np.abs(n(a[k]) - n(b[k])).max() <= 3 * 2.2 * c["lr"], k
                continue
            # atomics (fp64 statistics, fp32 wgrad accumulation) make sums order-dependent in the last
            # bits; Adam turns that into +-lr moves on noise-gradient elements only
            H.assert_adam_close(n(b[k]), n(a[k]), c["lr"], k, steps=3, frac=5e-2)
    np.testing.assert_allclose(eng.scalars(), eng2.scalars(), rtol=1e-4)
    assert eng2.adam_step == 3


def test_state_dict_roundtrip_and_reference_keys():
    import json
    man = json.load(open(os.path.join(G, "manifest.json")))
    cfg = planner.ModelCfg(kind="unimodal", z_dim=10, output_size=50)
    eng = Engine(cfg, 4)
    sd = eng.state_dict()
    assert sorted(sd.keys()) == sorted(k for k, _, _ in man["unimodal_z10_o50"])
    for k, shp, dt in man["unimodal_z10_o50"]:
        assert list(sd[k].shape) == shp, k
    om = O.OracleModel("unimodal", 10, 50, salt=1)
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    back = eng.state_dict()
    for k, v in om.state.items():
        np.testing.assert_array_equal(n(back[k]), n(v), err_msg=k)
    with pytest.raises(ValueError):
        bad = dict(back)
        bad["encoder.conv1.weight"] = torch.zeros(64, 1, 5)
        eng.load_state_dict(bad)
    cfgm = planner.ModelCfg(kind="multimodal", z_dim=10, output_size=50, output_size2=100)
    engm = Engine(cfgm, 4)
    assert sorted(engm.state_dict().keys()) == sorted(k for k, _, _ in man["multimodal_z10_o50_100"])


@pytest.mark.parametrize("B,L", [(2, 50), (3, 100), (65, 33), (513, 50), (6, 2), (7, 3), (64, 1)])
def test_ragged_batches_and_lengths(B, L):
    """Tiny, odd and tile-straddling batches / lengths (down to inputs of 1-3 samples, where every encoder stage is one position
    long and the stride-2 input-gradient has no odd phase): forward, loss and one full step against the oracle."""
    c = dict(kind="unimodal", z=10, L=L, B=B, with_class=False, beta=1.0, clip=1.0, lr=1e-4, salt=40 + B)
    eng, oms, batch, batch64, eps = build(**c)
    check_forward(eng, oms, batch, batch64, eps, training=False)
    outs64 = oms[1].forward(batch64, eps.double(), True)
    ls64 = oms[1].losses(batch64, outs64, 1.0)
    sc = eng.train_step()
    torch.cuda.synchronize()
    got = eng.scalars()
    np.testing.assert_allclose([got[0], got[1], got[3]], [float(v) for v in ls64], rtol=2e-4)
    sd = eng.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype.is_floating_point)
    assert eng.adam_step == 1


def _assert_same_units(a, b, eng, what="eval output of a unit depends on its batch"):
    """A unit's eval outputs from two batches.  On the fp32 matrix cores one conv body serves every batch size: bit for bit.  On the default
    three-term path a layer is served by the 64x64 body or, from two 128-row tiles per CU on, by a 128-row body — the same exact products summed
    in a different order — so batches of DIFFERENT size agree to fp32 rounding (measured 1-3e-6 of the tensor's max; bound 2e-5), and batches of
    the SAME size (the permutation checks next to the callers) bit for bit."""
    if eng.train_cfg.mfma_dtype == "f32":
        assert torch.equal(a, b), what
    else:
        err = float((a.double() - b.double()).abs().max() / a.double().abs().max().clamp_min(1e-30))
        assert err <= 2e-5, f"{what}: {err:.2e}"


def test_eval_rows_are_independent_of_batch_composition_at_config3_size():
    """BASELINE config 3's shape (batch 4096, L=256, z=32) is too large for the CPU oracle in a test, but the
    eval forward has a size-independent property: a unit's outputs do not depend on which other units share
    its batch.  The full batch must reproduce four quarter batches (bit for bit where the same kernels serve both: _assert_same_units),
    and the same batch with its units permuted must reproduce itself permuted, bit for bit."""
    B, L, z = 4096, 256, 32
    torch.manual_seed(0)
    cfg = planner.ModelCfg("unimodal", z, L)
    big = Engine(cfg, B)
    om = O.OracleModel("unimodal", z, L, salt=11)
    big.load_state_dict({k: v.detach() for k, v in om.state.items()})
    small = Engine(cfg, B // 4, share_params_from=big)
    x = torch.randn(B, 1, L, device="cuda")
    src = torch.randint(1, 5, (B,), device="cuda")
    eps = torch.randn(B, z, device="cuda")
    big.set_inputs(x, src, None, eps)
    full = [t.clone() for t in big.forward(False)]
    for q in range(4):
        sl = slice(q * B // 4, (q + 1) * B // 4)
        small.set_inputs(x[sl], src[sl], None, eps[sl])
        part = small.forward(False)
        for a, b in zip(full, part):
            _assert_same_units(a[sl], b, big)
    assert all(torch.isfinite(t).all() for t in full)
    perm = torch.randperm(B, device="cuda")
    big.set_inputs(x[perm], src[perm], None, eps[perm])
    for a, b in zip(full, big.forward(False)):
        assert torch.equal(a[perm], b), "eval output of a unit depends on its position in the batch"


def test_train_forward_is_permutation_equivariant_at_full_batch():
    """Batch 512 (the headline size), train mode: permuting the units permutes the outputs and leaves the
    loss scalars and the BatchNorm running statistics unchanged (up to the order of the fp64 statistic sums)."""
    B, L, z = 512, 100, 10
    eng = Engine(planner.ModelCfg("unimodal", z, L), B)
    om = O.OracleModel("unimodal", z, L, salt=12)
    sd = {k: v.detach() for k, v in om.state.items()}
    x, src, _, eps = O.synth_inputs(B, L, z, salt=12)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    res = []
    for p in (torch.arange(B), perm):
        eng.load_state_dict(sd)
        eng.set_inputs(x[p].cuda(), src[p].cuda(), None, eps[p].cuda())
        outs = [t.clone().cpu() for t in eng.forward(True)]
        res.append((outs, eng.scalars(), {k: v.cpu() for k, v in eng.state_dict().items() if "running" in k}))
    (o0, s0, r0), (o1, s1, r1) = res
    for a, b in zip(o0, o1):
        scale = float(a.abs().max())
        assert float((a[perm] - b).abs().max()) <= 2e-6 * max(scale, 1.0)
    np.testing.assert_allclose(s0, s1, rtol=1e-6)
    for k in r0:
        np.testing.assert_allclose(r0[k].numpy(), r1[k].numpy(), rtol=1e-6, atol=1e-7, err_msg=k)


def test_grouped_atomic_wgrad_equals_ordered_slab_reduction_at_full_batch():
    """Batch 512: the default weight-gradient path (one grouped launch, fp32 atomics, longest blocks first) against
    the deterministic one (per-conv launches writing slabs, reduced in order): two independent implementations of
    the same sums at the headline size."""
    B, L, z = 512, 100, 10
    om = O.OracleModel("unimodal", z, L, salt=13)
    sd = {k: v.detach() for k, v in om.state.items()}
    x, src, _, eps = O.synth_inputs(B, L, z, salt=13)
    grads = []
    for det in (False, True):
        eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(deterministic_wgrad=det))
        eng.load_state_dict(sd)
        eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
        eng.forward(True)
        eng.backward()
        torch.cuda.synchronize()
        grads.append({k: v.cpu().double() for k, v in eng.grad_dict().items()})
    for k, g in grads[1].items():
        if re.search(H.ZERO_GRAD_RE, k) or float(g.abs().max()) == 0.0:
            continue
        err = float((grads[0][k] - g).abs().max()) / float(g.abs().max())
        assert err <= 5e-5, (k, err)


@pytest.mark.parametrize("B,L", [(1, 50), (1, 5), (7, 2), (3, 1)])
def test_eval_forward_extreme_shapes(B, L):
    """Single-unit batches (legal in eval mode; train mode raises like torch's BatchNorm) and inputs so short that
    the stride-2 stages bottom out at length 1."""
    z = 10
    eng, oms, batch, batch64, eps = build("unimodal", z, L, B, False, 1.0, 0.0, 1e-3, 50 + L)
    check_forward(eng, oms, batch, batch64, eps, training=False)


def _rand_state(eng, seed):
    """random but well-conditioned parameters (torch default init scale) without the CPU oracle: the full-size
    property tests below compare the engine with itself"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    bn = set(eng.plan.bn_keys)
    for k, info in eng.plan.params.items():
        pre = k.rsplit(".", 1)[0]
        if k.endswith("embedding.weight"):
            v = torch.randn(info.shape, generator=g)
        elif pre in bn:
            v = 1.0 + 0.2 * (torch.rand(info.shape, generator=g) - 0.5) if k.endswith("weight") else 0.1 * (torch.rand(info.shape, generator=g) - 0.5)
        else:
            fan = 1
            for d_ in eng.plan.params[pre + ".weight"].shape[1:]:
                fan *= d_
            v = (torch.rand(info.shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5
        sd[k] = v
    eng.load_state_dict(sd, strict=False)
    return sd


def test_config5_multimodal_full_size_properties():
    """BASELINE config 5's per-rank shape — MultiModalCVAE, z=64, wave 256 + time 32, batch 8192 — is far too large
    for the CPU oracle inside a test (its tiny-batch twin is the `multi_c5` case above, pinned by a fixture from the
    reference).  At full size the engine is held to properties that do not depend on the size:
      * eval: a unit's outputs do not depend on which units share its batch (full batch == 4 quarter batches, bitwise);
      * train: permuting the units permutes the outputs and leaves losses / running statistics unchanged;
      * the default weight-gradient path (one grouped launch, fp32 atomics) equals the ordered slab reduction;
      * one full optimisation step moves every parameter tensor that has a gradient and keeps everything finite."""
    B, Lw, Lt, z = 8192, 256, 32, 64
    cfg = planner.ModelCfg("multimodal", z, Lw, Lt)
    gen = torch.Generator(device="cuda").manual_seed(5)
    x1 = torch.randn(B, 1, Lw, device="cuda", generator=gen)
    x2 = torch.rand(B, 1, Lt, device="cuda", generator=gen)
    src = torch.randint(1, 5, (B,), device="cuda", generator=gen)
    eps = torch.randn(B, z, device="cuda", generator=gen)
    big = Engine(cfg, B, planner.TrainCfg(lr=1e-3, clip=1.0))
    sd = _rand_state(big, 5)
    # -- eval: batch independence
    small = Engine(cfg, B // 4, share_params_from=big)
    big.set_inputs(x1, src, None, eps, x2=x2)
    full = [t.clone() for t in big.forward(False)]
    assert all(torch.isfinite(t).all() for t in full)
    for q in range(4):
        sl = slice(q * B // 4, (q + 1) * B // 4)
        small.set_inputs(x1[sl], src[sl], None, eps[sl], x2=x2[sl])
        for a, b in zip(full, small.forward(False)):
            _assert_same_units(a[sl], b, big)
    del small
    # -- train: permutation equivariance + gradients
    perm = torch.randperm(B, device="cuda", generator=gen)
    res = []
    for p_ in (torch.arange(B, device="cuda"), perm):
        big.load_state_dict(sd, strict=False)
        for k, info in big.plan.bufs.items():
            big.bufs[info.offset: info.offset + info.numel] = 1.0 if k.endswith("running_var") else 0.0
        big.set_inputs(x1[p_], src[p_], None, eps[p_], x2=x2[p_])
        outs = [t.clone() for t in big.forward(True)]
        big.backward()
        torch.cuda.synchronize()
        res.append((outs, big.scalars(), big.bufs.clone(), big.grads.clone()))
    (o0, s0, r0, g0), (o1, s1, r1, g1) = res
    for a, b in zip(o0, o1):
        assert float((a[perm] - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1.0)
    np.testing.assert_allclose(s0, s1, rtol=1e-5)
    assert float((r0 - r1).abs().max()) <= 1e-5 * float(r0.abs().max())
    # gradients are sums over the batch: a permutation only reorders the sums (and may flip a few leaky-ReLU branches)
    assert float((g0 - g1).norm() / g0.norm()) <= 2e-3
    # -- grouped atomic wgrad vs ordered slabs at full size
    det = Engine(cfg, B, planner.TrainCfg(lr=1e-3, clip=1.0, deterministic_wgrad=True), share_params_from=big)
    det.set_inputs(x1[perm], src[perm], None, eps[perm], x2=x2[perm])
    det.forward(True)
    det.backward()
    torch.cuda.synchronize()
    gd = det.grads        # shared arena: the deterministic pass overwrote it; compare with the saved atomic-path copy
    for k, info in big.plan.params.items():
        a, b = g1[info.offset: info.offset + info.numel], gd[info.offset: info.offset + info.numel]
        if float(b.abs().max()) == 0.0 or re.search(H.ZERO_GRAD_RE, k):
            continue
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()), k
    del det
    # -- one optimisation step
    before = big.params.clone()
    big.optimizer_step()
    torch.cuda.synchronize()
    assert big.adam_step == 1 and torch.isfinite(big.params).all()
    for k, info in big.plan.params.items():
        if k == "class_embedding.weight":
            continue
        sl = slice(info.offset, info.offset + info.numel)
        assert not torch.equal(before[sl], big.params[sl]), k


def test_config3_training_properties_at_batch_4096():
    """BASELINE config 3's shape (unimodal, z=32, wave 256 / time 32, batch 4096) in TRAINING mode: permutation
    equivariance of outputs, invariance of the losses, running statistics and gradients, and graph replay == eager."""
    B, z = 4096, 32
    gen = torch.Generator(device="cuda").manual_seed(3)
    for L in (256, 32):
        eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=1e-3, clip=1.0))
        sd = _rand_state(eng, L)
        x = torch.randn(B, 1, L, device="cuda", generator=gen)
        src = torch.randint(1, 5, (B,), device="cuda", generator=gen)
        eps = torch.randn(B, z, device="cuda", generator=gen)
        perm = torch.randperm(B, device="cuda", generator=gen)
        res = []
        for p_, use_graph in ((torch.arange(B, device="cuda"), False), (perm, False), (perm, True)):
            eng.load_state_dict(sd, strict=False)
            for k, info in eng.plan.bufs.items():
                eng.bufs[info.offset: info.offset + info.numel] = 1.0 if k.endswith("running_var") else 0.0
            eng.set_inputs(x[p_], src[p_], None, eps[p_])
            outs = [t.clone() for t in eng.forward(True, use_graph)]
            eng.backward(use_graph)
            torch.cuda.synchronize()
            res.append((outs, eng.scalars(), eng.bufs.clone(), eng.grads.clone()))
        (o0, s0, r0, g0), (o1, s1, r1, g1), (o2, s2, r2, g2) = res
        for a, b, c_ in zip(o0, o1, o2):
            assert float((a[perm] - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1.0)
            assert float((b - c_).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1.0)
        np.testing.assert_allclose(s0, s1, rtol=1e-5)
        np.testing.assert_allclose(s1, s2, rtol=1e-5)
        assert float((r0 - r1).abs().max()) <= 1e-5 * float(r0.abs().max())
        assert float((g0 - g1).norm() / g0.norm()) <= 2e-3
        assert float((g1 - g2).norm() / g1.norm()) <= 2e-3
        assert torch.isfinite(g0).all()
        del eng


def test_staged_step_is_one_graph_and_equals_the_host_staged_step():
    """TrainCfg(resident_units=N): "step_staged" (HP_OP_STAGE_BATCH + forward + backward + optimiser, ONE hipGraph replay per
    optimisation step, no torch kernel in it) against an engine that is handed the same rows and the same Philox noise by the
    host: same losses and parameters, step after step, through a wrap of the permutation."""
    from oracle import interp
    z, L, B, N = 10, 50, 16, 80
    cfg = planner.ModelCfg("unimodal", z, L)
    om = O.OracleModel("unimodal", z, L, salt=17)
    x, src, cls, _ = O.synth_inputs(N, L, z, salt=17)
    tc = planner.TrainCfg(lr=1e-5, clip=1.0, deterministic_wgrad=True)
    staged = Engine(cfg, B, dataclasses.replace(tc, resident_units=N))
    plain = Engine(cfg, B, tc)
    for e in (staged, plain):
        e.load_state_dict({k: v.detach() for k, v in om.state.items()})
    perm = torch.from_numpy(np.random.default_rng(3).permutation(N))
    staged.load_dataset(x.cuda().reshape(N, L), src.cuda(), perm=perm.cuda(), seed=99)
    spe = N // B
    for step in range(spe + 2):
        staged.train_step_staged(use_graph=True)
        rows = perm[(step % spe) * B: (step % spe + 1) * B]
        # what the in-graph staging put into the input slots: the permutation's rows, exactly, and the Philox stream of (seed, step)
        xs, ss, es = staged.io("x").clone(), staged.io("src").clone(), staged.io("eps").clone()
        assert torch.equal(xs.cpu().reshape(B, L), x[rows].reshape(B, L)) and torch.equal(ss.cpu(), src[rows])
        assert (es.cpu() - torch.from_numpy(interp.philox_normal(99, step, B * z).reshape(B, z))).abs().max() <= 5e-6
        # the host-staged engine on the very same inputs (the device's float32 log / sincos differ from numpy's in the last bits, and
        # Adam would amplify that over the steps)
        plain.set_inputs(xs, ss, None, es)
        plain.train_step(use_graph=True)
        torch.cuda.synchronize()
        np.testing.assert_allclose(staged.scalars(), plain.scalars(), rtol=2e-6, err_msg=f"step {step}")
    assert int(staged.io("cursor")[0]) == spe + 2 and staged.adam_step == plain.adam_step == spe + 2
    a, b = staged.state_dict(), plain.state_dict()
    for k in a:
        if a[k].dtype.is_floating_point:
            H.assert_adam_close(n(a[k]), n(b[k]), 1e-5, k, steps=spe + 2, frac=2e-2)
    with pytest.raises(Exception):
        plain.train_step_staged()


@pytest.mark.parametrize("L,clip,steps,bf16", [(50, None, 160, False), (100, 1.0, 120, False), (50, None, 160, True)])
def test_long_run_loss_curves_track_the_oracle(L, clip, steps, bf16):
    """160 optimisation steps of the wave cVAE on the same batches and noise in the engine and in the CPU oracle: single trajectories
    diverge chaotically, the loss CURVES must not (mse within a factor of 2.5 and loss within 3 at every checkpoint, plateau within [0.6, 1.67]).
    Guards what single-step parity cannot see: bias correction at large step counts, weight decay, the KL weight, running statistics
    (tools/long_run_vs_oracle.py; at 1 500 steps the plateaus agree to 1 %)."""
    from tools import long_run_vs_oracle as lr
    # (the second case: the time model's shape, with gradient clipping; the third: bf16 matrix path with bf16-STORED activations — the
    #  mode must train, not merely step: same curve criteria against the fp32 CPU oracle)
    rows = lr.run(steps=steps, B=128, L=L, clip=clip, verbose=False, **(dict(mfma_dtype="bf16", act_dtype="bf16") if bf16 else {}))
    bad, tail_e, tail_o = lr.check(rows)
    assert not bad, bad
    assert 0.6 <= tail_e / tail_o <= 1.67, (tail_e, tail_o)
    assert rows[-1][1][0] < 0.05 * rows[0][1][0]          # and the loss really fell (wave: 8.6 -> < 0.1)
