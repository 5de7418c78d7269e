"""GPU: the fused heads launch (HP_OP_HEADS, csrc/heads_fused.h — one workgroup for the 11 training-forward head ops between the
backbones) against the SAME program with the chain left as individual launches (TrainCfg(fuse_heads=False)), and the plumbing around the
launch unit (range checks, refusal of a chain that is not the heads chain)."""
import numpy as np
import pytest
import torch

from hippie_amd import planner, program as P
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O

pytestmark = pytest.mark.gpu


def _engine(z, L, B, with_class, fuse, clip=1.0):
    cfg = planner.ModelCfg(kind="unimodal", z_dim=z, output_size=L)
    # (no liveness packing: the test reads backward-pass tensors after the pass, when packed memory has long been reused)
    eng = Engine(cfg, B, planner.TrainCfg(lr=1e-3, clip=clip, fuse_heads=fuse, deterministic_wgrad=True, reuse_workspace=False), with_class=with_class)
    om = O.OracleModel("unimodal", z, L, salt=11)
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=11)
    eng.set_inputs(x.cuda(), src.cuda(), cls.cuda() if with_class else None, eps.cuda())
    return eng


def _heads(plan):
    return [(k, [int(v) for v in r["i"][:3]]) for k, r in enumerate(plan.ops.recs) if int(r["op"]) == P.HEADS]


@pytest.mark.parametrize("z,L,B,with_class", [(10, 50, 512, False), (10, 100, 512, True), (10, 50, 37, True), (5, 50, 64, False), (10, 50, 1, False)])
def test_fused_heads_equal_the_unfused_chain(z, L, B, with_class):
    """Same arithmetic per element; only the order of the fp64 BatchNorm / KL sums differs (a fixed tree in the fused kernel, atomics in
    the generic ones): every tensor the chain writes, every gradient, the scalars and the parameters after two optimisation steps agree to
    1e-6 of the tensor's scale (most are bit-equal)."""
    fused, plain = _engine(z, L, B, with_class, True), _engine(z, L, B, with_class, False)
    hf = _heads(fused.plan)
    assert [h[1][2] for h in hf] == [0] and [h[1][1] for h in hf] == [11] and not _heads(plain.plan)
    launches = lambda pl: sum(1 for r in pl.ops.recs if not int(r["flags"]) & P.FLAG_MEMBER)
    assert launches(plain.plan) - launches(fused.plan) == 10
    # the chains' tensors live at the same workspace offsets in both lowerings only without liveness packing; compare by member record
    def tensors(eng, first, count):
        out = {}
        for k in range(first, first + count):
            r = eng.plan.ops.recs[k]
            opc = int(r["op"])
            slot = {P.CONCAT: 0, P.LINEAR_FWD: 3, P.BN_APPLY: 1, P.REPARAM_KL_FWD: 2, P.BN_BWD_REDUCE: 3, P.BN_BWD_APPLY: 5, P.LINEAR_BWD_X: 2, P.REPARAM_KL_BWD: 3}[opc]
            n_ = {P.CONCAT: int(r["i"][0]) * int(r["i"][2]), P.LINEAR_FWD: int(r["i"][0]) * int(r["i"][4]), P.BN_APPLY: int(r["i"][0]) * int(r["i"][1]),
                  P.REPARAM_KL_FWD: int(r["i"][0]) * int(r["i"][1]), P.BN_BWD_REDUCE: int(r["i"][0]) * int(r["i"][1]), P.BN_BWD_APPLY: int(r["i"][0]) * int(r["i"][1]),
                  P.LINEAR_BWD_X: int(r["i"][0]) * int(r["i"][4]), P.REPARAM_KL_BWD: int(r["i"][0]) * 2 * int(r["i"][1])}[opc]
            ref = int(r["buf"][slot])
            assert (ref >> 56) == P.WS
            off = ref & ((1 << 56) - 1)
            out[f"{k - first}:{P.OP_NAMES[opc]}"] = eng.ws[off: off + 4 * n_].view(torch.float32).clone()
        return out
    for step in range(2):
        for e in (fused, plain):
            e.forward(True, use_graph=(step == 1))
        torch.cuda.synchronize()
        tf = tensors(fused, hf[0][1][0], 11)
        plain_first = next(k for k, r in enumerate(plain.plan.ops.recs) if int(r["op"]) == P.CONCAT)
        tp = tensors(plain, plain_first, 11)
        for (ka, a), (kb, b) in zip(tf.items(), tp.items()):
            assert ka == kb
            err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
            assert err <= 1e-6, f"step {step} forward member {ka}: {err:.2e}"
        np.testing.assert_allclose(fused.scalars(), plain.scalars(), rtol=1e-6)
        for e in (fused, plain):
            e.backward(use_graph=(step == 1))
        torch.cuda.synchronize()
        gf, gp = fused.grad_dict(), plain.grad_dict()
        for k in gf:
            err = float((gf[k] - gp[k]).abs().max()) / max(float(gp[k].abs().max()), 1e-30)
            assert err <= 2e-6, f"step {step} grad {k}: {err:.2e}"
        for e in (fused, plain):
            e.optimizer_step(use_graph=(step == 1))
    torch.cuda.synchronize()
    sf, sp = fused.state_dict(), plain.state_dict()
    for k in sf:
        if sf[k].dtype.is_floating_point:
            # (Adam turns a last-bit gradient difference into at most a +-lr step: bounded by 2.2 lr per step, almost everywhere far less)
            assert float((sf[k] - sp[k]).abs().max()) <= 2 * 2.2e-3 + 1e-6 * float(sp[k].abs().max()), k
    n_diff = sum(int((sf[k] != sp[k]).sum()) for k in sf if sf[k].dtype.is_floating_point)
    n_all = sum(sf[k].numel() for k in sf if sf[k].dtype.is_floating_point)
    assert n_diff <= 0.02 * n_all, f"{n_diff} of {n_all} parameter / buffer elements differ after two steps"


def test_fused_heads_are_not_emitted_where_the_kernel_does_not_apply():
    lower = lambda **kw: planner.lower(planner.ModelCfg(kind=kw.pop("kind", "unimodal"), z_dim=kw.pop("z", 10), output_size=50, output_size2=100),
                                       kw.pop("B", 64), planner.TrainCfg(**kw))
    assert len(_heads(lower())) == 1
    assert not _heads(lower(z=32)) and not _heads(lower(B=513)) and not _heads(lower(kind="multimodal")) and not _heads(lower(fuse_heads=False))
    assert not _heads(lower(sync_bn_world=2)) and not _heads(lower(group_small_wgrads=False))


def test_heads_unit_range_checks_and_refusal_of_other_chains():
    plan = planner.lower(planner.ModelCfg(kind="unimodal", z_dim=10, output_size=50), 16, planner.TrainCfg())
    ((kf, (first, count, _)),) = _heads(plan)
    eng = Engine(planner.ModelCfg(kind="unimodal", z_dim=10, output_size=50), 16, planner.TrainCfg())
    with pytest.raises(P.HipEngineError):
        eng.prog.run(first + 2, count)                       # cuts through the unit
    with pytest.raises(P.HipEngineError):
        eng.prog.run(kf, 1)                                  # the closing record without its members
    ops = plan.ops.array().copy()
    ops[first + 1]["i"][1] += 1                              # a member that is no longer the chain's Linear
    n = plan.n_param_floats * 4
    dev = [torch.zeros(max(s, 8), dtype=torch.uint8, device="cuda") for s in (plan.ws_bytes, n, n, plan.n_buf_floats * 4, n, n)]
    with pytest.raises(P.HipEngineError, match="heads"):
        P.DeviceProgram(ops, [d.data_ptr() for d in dev], [d.numel() for d in dev])
