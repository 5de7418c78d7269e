"""f4: Schedule-Free AdamW (hippie/optimizers.py:18-209).

CPU: the oracle restatement and the op semantics (oracle/interp.py) against vectors produced by the
reference class itself (tests/golden/schedulefree_*.npz, generator make_golden_optim.py).
GPU: the HIP kernels through the C ABI against the same vectors, and the host class on a model.
Tolerance: fp32 elementwise arithmetic, rtol 2e-6 / atol 1e-7 per step (fma contraction only).
"""
import os

import numpy as np
import pytest
import torch

from hippie_amd import planner, program as P
from hippie_amd.program import Ref
from oracle import interp
from oracle.optim_oracle import ScheduleFreeOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "default": dict(),
    "warmup_decay": dict(lr=1e-2, weight_decay=0.01, warmup_steps=3, r=0.5, weight_lr_power=2.0),
    "loop_path": dict(lr=5e-3, betas=(0.8, 0.99), weight_decay=0.1, warmup_steps=2),
}
NT = 4      # tensors per case


def load(name):
    return np.load(os.path.join(GOLD, f"schedulefree_{name}.npz"))


def cat(g, prefix):
    return np.concatenate([g[f"{prefix}_{j}"].reshape(-1) for j in range(NT)])


def close(got, want, what, rtol=2e-6, atol=1e-7):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_reference_vectors(name):
    g = load(name)
    params = {j: torch.tensor(g[f"p0_{j}"]) for j in range(NT)}
    opt = ScheduleFreeOracle(params, **CASES[name])
    swap = int(g["swap_after"])
    for t in range(int(g["n_steps"])):
        opt.step({j: torch.tensor(g[f"g{t}_{j}"]) for j in range(NT)})
        np.testing.assert_allclose([opt.k, opt.weight_sum, opt.lr_max], g[f"group{t}"], rtol=1e-14)
        for j in range(NT):
            close(params[j].numpy(), g[f"y{t}_{j}"], f"y step {t}")
            close(opt.z[j].numpy(), g[f"z{t}_{j}"], f"z step {t}")
            close(opt.exp_avg_sq[j].numpy(), g[f"v{t}_{j}"], f"v step {t}", rtol=1e-6, atol=0)
        if t == swap:
            opt.eval()
            for j in range(NT):
                close(params[j].numpy(), g[f"x{t}_{j}"], "x (eval point)")
            assert int(g["eval_step_raises"]) == 1
            with pytest.raises(Exception, match="Not in train mode"):
                opt.step({j: torch.tensor(g[f"g{t}_{j}"]) for j in range(NT)})
            np.testing.assert_allclose([opt.k, opt.weight_sum, opt.lr_max], g["group_after_raise"], rtol=1e-14)
            opt.train()
            for j in range(NT):
                close(params[j].numpy(), g[f"yback{t}_{j}"], "y after eval/train round trip")


def sf_program(g, kw, n):
    """A flat-vector program: for each step [SF_SCHEDULE, ADAMW_SF, STEP_INC] (+ the eval/raise/train sequence)."""
    b1, b2 = kw.get("betas", (0.9, 0.999))
    off = 0

    def put(nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        r = Ref(P.WS, off)
        off += nbytes
        return r
    steps, swap = int(g["n_steps"]), int(g["swap_after"])
    y, z, v = put(4 * n), put(4 * n), put(4 * n)
    grads = [put(4 * n) for _ in range(steps)]
    snaps = {}
    step, st, norm2 = put(8), put(32), put(8)
    ol = P.OpList()
    sched = dict(i=[kw.get("warmup_steps", 0)], f=[kw.get("lr", 0.0025), 1.0 - b2, kw.get("r", 0.0), kw.get("weight_lr_power", 2.0)],
                 buf=[step, st])
    for t in range(steps):
        ol.add(P.SF_SCHEDULE, 0, **sched)
        ol.add(P.ADAMW_SF, 0, i=[n], f=[b1, b2, kw.get("eps", 1e-8), kw.get("weight_decay", 0.0), 0.0, 1.0 - b2],
               buf=[y, grads[t], z, v, step, st, norm2])
        ol.add(P.STEP_INC, 0, buf=[step])
        if t == swap:
            ol.add(P.LERP, 0, i=[n], f=[1.0 - 1.0 / b1], buf=[y, z])          # eval()
            ol.add(P.SF_SCHEDULE, 0, **sched)                                   # step() in eval mode: raises after this
            ol.add(P.LERP, 0, i=[n], f=[1.0 - b1], buf=[y, z])                # train()
    image = np.zeros(off + 256, np.uint8)
    image[y.offset: y.offset + 4 * n] = cat(g, "p0").view(np.uint8)
    for t in range(steps):
        image[grads[t].offset: grads[t].offset + 4 * n] = cat(g, f"g{t}").view(np.uint8)
    return ol.array(), image, dict(y=y, z=z, v=v, step=step, st=st)


def f32(mem, ref, n):
    return mem[ref.offset: ref.offset + 4 * n].view(np.float32)


def check_final(g, mem, refs, n, name):
    T = int(g["n_steps"]) - 1
    # error compounds over the steps: a few ulp of the parameter scale
    close(f32(mem, refs["y"], n), cat(g, f"y{T}"), name + " y", rtol=1e-5, atol=1e-6)
    close(f32(mem, refs["z"], n), cat(g, f"z{T}"), name + " z", rtol=1e-5, atol=1e-6)
    close(f32(mem, refs["v"], n), cat(g, f"v{T}"), name + " v", rtol=1e-5, atol=0)
    grp = g[f"group{T}"]
    st = mem[refs["st"].offset: refs["st"].offset + 32].view(np.float64)
    assert mem[refs["step"].offset: refs["step"].offset + 8].view(np.int64)[0] == int(grp[0])
    # f[] carries lr / beta2 as fp32, so the fp64 schedule differs from the reference's at 1e-7 relative
    np.testing.assert_allclose([st[1], st[0]], grp[1:], rtol=1e-6)


@pytest.mark.parametrize("name", list(CASES))
def test_op_semantics_reproduce_reference_vectors(name):
    g = load(name)
    n = cat(g, "p0").size
    recs, image, refs = sf_program(g, CASES[name], n)
    A = interp.Arenas([image.size, 4, 4, 4, 4, 4])
    A.mem[0][:] = image
    interp.run(recs, A)
    check_final(g, A.mem[0], refs, n, name)


def test_planner_lowers_schedulefree_segments():
    cfg = planner.ModelCfg("unimodal", 10, 50)
    plan = planner.lower(cfg, 4, planner.TrainCfg(optimizer="schedulefree", warmup_steps=5, clip=1.0))
    first, count = plan.ops.segments["opt"]
    ops = [int(plan.ops.array()[k]["op"]) for k in range(first, first + count)]
    assert ops == [P.ZERO, P.GRADNORM, P.SF_SCHEDULE, P.ADAMW_SF, P.STEP_INC]       # ZERO: the norm accumulator, per optimiser step
    for seg in ("sf_eval", "sf_train"):
        f, c = plan.ops.segments[seg]
        assert c == 1 and int(plan.ops.array()[f]["op"]) == P.LERP
    adam = planner.lower(cfg, 4, planner.TrainCfg())
    assert plan.n_buf_floats == adam.n_buf_floats and "sf_eval" not in adam.ops.segments
    with pytest.raises(ValueError):
        planner.lower(cfg, 4, planner.TrainCfg(optimizer="sgd"))


# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_kernels_reproduce_reference_vectors(name):
    g = load(name)
    n = cat(g, "p0").size
    recs, image, refs = sf_program(g, CASES[name], n)
    dev = torch.from_numpy(image.copy()).cuda()
    bases = [dev.data_ptr()] + [0] * 5
    for r in recs:
        P.run_single_op(r, bases, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    check_final(g, dev.cpu().numpy(), refs, n, name)


@pytest.mark.gpu
def test_gpu_host_class_trains_and_swaps_like_the_oracle():
    """hippie_amd.optimizers.AdamWScheduleFree on a model: 3 steps against OracleModel grads + ScheduleFreeOracle
    (fp64), eval()/train() swap, 'Not in train mode!' and checkpoint round trip."""
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    from hippie_amd.optimizers import AdamWScheduleFree
    from oracle import cvae_oracle as O
    from tests.helpers import ZERO_GRAD_RE, assert_adam_close
    import re
    z, L, B = 10, 50, 16
    kw = dict(lr=1e-6, weight_decay=0.01, warmup_steps=2)
    om = O.OracleModel("unimodal", z, L, dtype=torch.float64, salt=5)
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    net.load_state_dict({k: v.detach().float() for k, v in om.state.items()}, strict=False)
    module = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3, weight_decay=0.0)
    module.optimizer = opt = AdamWScheduleFree(net.parameters(), **kw)
    p0 = {k: om.state[k].detach().clone() for k in om.param_keys}
    params = {k: om.state[k] for k in om.param_keys if k != "class_embedding.weight"}
    oo = ScheduleFreeOracle({k: v.detach() for k, v in params.items()}, **kw)
    for t in range(3):
        x, src, _, eps = O.synth_inputs(B, L, z, salt=20 + t)
        for k in om.param_keys:
            om.state[k].grad = None
        outs = om.forward((x.double(), src, None), eps.double(), True)
        om.losses((x.double(), src, None), outs)[0].backward()
        oo.step({k: om.state[k].grad for k in params})
        eng = net.engine(B, False)
        net.train()
        net._run_forward(eng, x.cuda(), src.cuda(), None, eps.cuda())
        opt.last_engine = eng
        eng.backward()
        opt.step()
    got = net.state_dict()
    assert opt.param_groups[0]["k"] == 3
    np.testing.assert_allclose([opt.param_groups[0]["weight_sum"], opt.param_groups[0]["lr_max"]], [oo.weight_sum, oo.lr_max], rtol=1e-6)
    for k in params:
        if re.fullmatch(ZERO_GRAD_RE, k):
            d_got = got[k].cpu().double() - p0[k]
            assert d_got.abs().max() <= 2.2 * 3 * kw["lr"] + 1e-9, k
            continue
        # the normalised step is ~lr*sign(g) early on, exactly like Adam: same criterion as the AdamW trajectories
        assert_adam_close(got[k].cpu().numpy(), om.state[k].detach().numpy(), kw["lr"], msg=k, steps=3, frac=5e-2)
    # eval(): parameters move to x = lerp(y, z, 1 - 1/beta1); train(): back to y
    y = {k: v.clone() for k, v in net.state_dict().items()}
    zs = opt.state_dict()["state"]
    module.eval()
    xs = net.state_dict()
    names = list(net._any_engine().plan.params)
    k0 = names[0]
    want = torch.lerp(y[k0], zs[0]["z"], 1 - 1 / 0.9)
    torch.testing.assert_close(xs[k0], want, rtol=1e-6, atol=1e-7)
    with pytest.raises(Exception, match="Not in train mode"):
        opt.step()
    module.train()
    back = net.state_dict()
    for k in params:
        torch.testing.assert_close(back[k], y[k], rtol=1e-5, atol=1e-6)
    # checkpoint round trip of the optimiser state
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"z", "exp_avg_sq"} and sd["param_groups"][0]["k"] == 3
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["k"] == 3
