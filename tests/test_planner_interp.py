"""Host-side lowering (hippie_amd.planner) executed by the numpy op interpreter must equal the
torch oracle: forward outputs, loss scalars, every gradient, one AdamW step, BN running stats."""
import re

import numpy as np
import pytest
import torch

from hippie_amd import planner, program as P
from oracle import cvae_oracle as O
from oracle import interp
from tests import helpers as H

torch.set_num_threads(4)


def run_case(kind, z, L, B, with_class, beta, clip, lr, salt, L2=None, w1=1.0, w2=1.0, deterministic=False):
    cfg = planner.ModelCfg(kind=kind, z_dim=z, output_size=L, output_size2=L2 or 100)
    tc = planner.TrainCfg(lr=lr, weight_decay=0.01, beta=beta, clip=clip or 0.0, w1=w1, w2=w2, deterministic_wgrad=deterministic)
    plan = planner.lower(cfg, B, tc, with_class=with_class)
    ops = plan.ops.array()
    A = H.make_arenas(plan)
    om = O.OracleModel(kind, z, L, output_size2=L2, salt=salt)
    H.load_state(plan, A, om.state)
    if kind == "unimodal":
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt)
        batch = (x, src, cls if with_class else None)
        H.set_io(plan, A, "x", x.numpy())
    else:
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt, name="x1")
        x2, _, _, _ = O.synth_inputs(B, L2, z, salt=salt, name="x2")
        batch = (x, x2, src, cls if with_class else None)
        H.set_io(plan, A, "x", x.numpy())
        H.set_io(plan, A, "x2", x2.numpy())
    H.set_io(plan, A, "src", src.numpy())
    H.set_io(plan, A, "cls", cls.numpy())
    H.set_io(plan, A, "eps", eps.numpy())

    # float64 twin of the oracle: the yardstick of the parity criterion (tests/helpers.parity)
    om64 = O.OracleModel(kind, z, L, output_size2=L2, salt=salt, dtype=torch.float64)
    batch64 = tuple(t.double() if (t is not None and t.is_floating_point()) else t for t in batch)
    eps64 = eps.double()

    # ---- eval forward (running statistics) ----
    s, c = plan.ops.segments["fwd_eval"]
    interp.run(ops, A, s, c)
    with torch.no_grad():
        outs = om.forward(batch, eps, training=False)
        outs64 = om64.forward(batch64, eps64, training=False)
    H.parity(H.get_io(plan, A, "enc_eval"), outs[0].numpy(), outs64[0].numpy(), "eval enc")
    H.parity(H.get_io(plan, A, "mulv_eval")[:, :z], outs[1].numpy(), outs64[1].numpy(), "eval mu")
    H.parity(H.get_io(plan, A, "rec_eval"), outs[3].numpy(), outs64[3].numpy(), "eval rec")

    # ---- one full training step ----
    for seg in ("fwd_train", "bwd"):
        s, c = plan.ops.segments[seg]
        interp.run(ops, A, s, c)
    outs = om.forward(batch, eps, True)
    ls = om.losses(batch, outs, beta, w1, w2)
    ls[0].backward()
    outs64 = om64.forward(batch64, eps64, True)
    ls64 = om64.losses(batch64, outs64, beta, w1, w2)
    ls64[0].backward()
    n = lambda t: t.detach().numpy()
    H.parity(H.get_io(plan, A, "enc_train"), n(outs[0]), n(outs64[0]), "enc")
    mulv = H.get_io(plan, A, "mulv_train")
    H.parity(mulv[:, :z], n(outs[1]), n(outs64[1]), "mu")
    H.parity(mulv[:, z:], n(outs[2]), n(outs64[2]), "logvar")
    H.parity(H.get_io(plan, A, "rec_train"), n(outs[3]), n(outs64[3]), "rec")
    sc = H.get_io(plan, A, "scalars")
    if kind == "unimodal":
        np.testing.assert_allclose(sc[[0, 1, 3]], [float(v) for v in ls64], rtol=1e-5)
    else:
        H.parity(H.get_io(plan, A, "rec2_train"), n(outs[4]), n(outs64[4]), "rec2")
        np.testing.assert_allclose(sc, [float(v) for v in ls64], rtol=1e-5)
    # gradients
    grads = H.read_params(plan, A, P.GRAD)
    og, og64 = om.grads(), om64.grads()
    for k, g in og.items():
        mine = grads[k]
        if g is None:
            assert np.all(mine == 0), k
            continue
        if re.search(H.ZERO_GRAD_RE, k):
            assert np.abs(g.numpy()).max() < 1e-5 and np.abs(mine).max() < 1e-5, k
            continue
        H.parity(mine, g.numpy(), og64[k].numpy(), "grad " + k)
    # running stats after the training forward
    bufs = H.read_bufs(plan, A)
    for k, v in bufs.items():
        np.testing.assert_allclose(v, om.state[k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    # optimiser (oracle: same grads -> clip -> AdamW)
    s, c = plan.ops.segments["opt"]
    interp.run(ops, A, s, c)
    with torch.no_grad():
        g = om.grads()
        if clip:
            O.clip_grad_norm(list(g.values()), clip)
        for k in om.param_keys:
            if g[k] is not None:
                om.exp_avg[k] = torch.zeros_like(om.state[k])
                om.exp_avg_sq[k] = torch.zeros_like(om.state[k])
        O.adamw_step({k: om.state[k] for k in om.param_keys}, g, om.exp_avg, om.exp_avg_sq, 1, lr, 0.01)
    params = H.read_params(plan, A)
    for k in om.param_keys:
        if re.search(H.ZERO_GRAD_RE, k):
            assert np.abs(params[k] - om.state[k].detach().numpy()).max() <= 2.2 * lr, k
            continue
        H.assert_adam_close(params[k], om.state[k].detach().numpy(), lr, k, grad=(g[k].numpy() if g[k] is not None else None))
    return plan


def test_unimodal_wave_step():
    plan = run_case("unimodal", 10, 50, 8, False, 1.0, None, 1e-3, 0)
    # parameter count matches the reference (8 056 639 incl. class_embedding)
    assert sum(i.numel for i in plan.params.values()) == 8056639
    assert plan.n_active <= plan.params["class_embedding.weight"].offset


def test_unimodal_time_step_with_clip_and_class_labels():
    run_case("unimodal", 5, 100, 6, True, 0.5, 1.0, 1e-3, 3)


def test_unimodal_odd_length():
    run_case("unimodal", 10, 32, 5, False, 1.0, None, 1e-3, 6)


def test_unimodal_deterministic_wgrad_slabs():
    plan = run_case("unimodal", 10, 50, 4, False, 1.0, None, 1e-3, 2, deterministic=True)
    assert any(int(r["op"]) == P.SLAB_REDUCE for r in plan.ops.recs)


def test_multimodal_step():
    run_case("multimodal", 10, 50, 6, False, 1.0, 1.0, 1e-3, 7, L2=100, w1=1.0, w2=0.5)


def test_forward_flop_count_matches_survey_probe():
    """SURVEY.md section 2.1 [probe]: conv+linear forward FLOPs per sample (2*MAC)."""
    for L, z, want in ((50, 10, 98585744), (100, 10, 131336976), (256, 32, 232528384), (32, 32, 80151040)):
        pl = planner.lower(planner.ModelCfg("unimodal", z, L), 4, planner.TrainCfg())
        assert pl.flops_fwd // 4 == want, (L, z, pl.flops_fwd // 4)


def run_lockstep(plan, ops, arenas, seg):
    """Data-parallel ranks in lock step on the interpreter: at every HP_OP_STATS_SYNC marker the slot is summed
    over the ranks' arenas — what Engine.run does with an all-reduce in sync-BatchNorm mode."""
    first, count = plan.ops.segments[seg]
    cur = first
    for k in range(first, first + count):
        if int(ops[k]["op"]) == P.STATS_SYNC:
            for A in arenas:
                interp.run(ops, A, cur, k - cur)
            n, ref = int(ops[k]["i"][0]), ops[k]["buf"][0]
            views = [A.f64(ref, n) for A in arenas]
            tot = np.sum(views, axis=0)
            for v in views:
                v[:] = tot
            cur = k + 1
    for A in arenas:
        interp.run(ops, A, cur, first + count - cur)


@pytest.mark.parametrize("kind", ["unimodal", "multimodal"])
def test_sync_batchnorm_two_ranks_equal_the_global_batch_oracle(kind):
    """TrainCfg(sync_bn_world=2): two ranks with B rows each must reproduce the single-process oracle at 2B rows
    (forward rows, running statistics, and mean-over-ranks gradients) — torch.nn.SyncBatchNorm + DDP semantics."""
    B, z, W = 4, 10, 2
    L2 = 100 if kind == "multimodal" else None
    cfg = planner.ModelCfg(kind=kind, z_dim=z, output_size=50, output_size2=L2 or 100)
    plan = planner.lower(cfg, B, planner.TrainCfg(sync_bn_world=W), with_class=True)
    ops = plan.ops.array()
    assert sum(int(r["op"]) == P.STATS_SYNC for r in ops) > 80
    om = O.OracleModel(kind, z, 50, output_size2=L2, salt=6)
    om64 = O.OracleModel(kind, z, 50, output_size2=L2, salt=6, dtype=torch.float64)
    x, src, cls, eps = O.synth_inputs(W * B, 50, z, salt=6, name="x1")
    x2 = O.synth_inputs(W * B, 100, z, salt=6, name="x2")[0]
    arenas = []
    for r in range(W):
        A = H.make_arenas(plan)
        H.load_state(plan, A, om.state)
        sl = slice(r * B, (r + 1) * B)
        H.set_io(plan, A, "x", x[sl].numpy())
        if kind == "multimodal":
            H.set_io(plan, A, "x2", x2[sl].numpy())
        for nm, v in (("src", src), ("cls", cls), ("eps", eps)):
            H.set_io(plan, A, nm, v[sl].numpy())
        arenas.append(A)
    run_lockstep(plan, ops, arenas, "fwd_train")
    run_lockstep(plan, ops, arenas, "bwd")
    batch = (x, src, cls) if kind == "unimodal" else (x, x2, src, cls)
    batch64 = tuple(t.double() if t.is_floating_point() else t for t in batch)
    outs = om.forward(batch, eps, True)
    om.losses(batch, outs)[0].backward()
    outs64 = om64.forward(batch64, eps.double(), True)
    om64.losses(batch64, outs64)[0].backward()
    n = lambda t: t.detach().numpy()
    for r, A in enumerate(arenas):
        sl = slice(r * B, (r + 1) * B)
        H.parity(H.get_io(plan, A, "enc_train"), n(outs[0])[sl], n(outs64[0])[sl], f"rank {r} enc")
        H.parity(H.get_io(plan, A, "rec_train"), n(outs[3])[sl], n(outs64[3])[sl], f"rank {r} rec")
    # running statistics: every rank holds the GLOBAL batch statistics (unbiased variance over W*M rows)
    for A in arenas:
        for k, v in H.read_bufs(plan, A).items():
            np.testing.assert_allclose(v, om.state[k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    grads = [H.read_params(plan, A, P.GRAD) for A in arenas]
    og, og64 = om.grads(), om64.grads()
    for k, g in og.items():
        mine = (grads[0][k] + grads[1][k]) / 2          # the data-parallel gradient mean
        if re.search(H.ZERO_GRAD_RE, k):
            assert np.abs(mine).max() < 1e-5, k
            continue
        H.parity(mine, g.numpy(), og64[k].numpy(), "sync-BN DP grad " + k)


def test_staged_step_gathers_the_batch_and_walks_the_permutation():
    """TrainCfg(resident_units=N): HP_OP_STAGE_BATCH + cursor increment in front of the training forward.  Through the
    interpreter: each staged step sees exactly the rows perm[j*B:(j+1)*B] (j = (cursor mod batches) * world + rank), labels
    follow their rows, eps is the Philox stream of (seed, cursor, rank), and "step_staged" is the contiguous range stage..opt."""
    B, z, L, N = 4, 10, 50, 19
    plan = planner.lower(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(resident_units=N, dp_world=2, dp_rank=1))
    ops = plan.ops.array()
    segs = plan.ops.segments
    assert segs["step_staged"] == (segs["stage"][0], segs["stage"][1] + segs["step"][1]) and sum(segs["stage"]) == segs["step"][0]
    om = O.OracleModel("unimodal", z, L, salt=2)
    x, src, cls, _ = O.synth_inputs(N, L, z, salt=2)
    A = H.make_arenas(plan)
    H.load_state(plan, A, om.state)
    H.set_io(plan, A, "data_x", x.numpy().reshape(N, L))
    H.set_io(plan, A, "data_labels", src.numpy())
    perm = np.random.default_rng(0).permutation(N)
    H.set_io(plan, A, "perm", perm)
    H.set_io(plan, A, "seed", [77])
    spe = N // (B * 2)
    for step in range(2 * spe + 1):                        # wraps around the permutation
        interp.run(ops, A, *segs["stage"])
        j = (step % spe) * 2 + 1
        rows = perm[j * B: (j + 1) * B]
        np.testing.assert_array_equal(H.get_io(plan, A, "x").reshape(B, L), x.numpy().reshape(N, L)[rows])
        np.testing.assert_array_equal(H.get_io(plan, A, "src"), src.numpy()[rows])
        np.testing.assert_array_equal(H.get_io(plan, A, "eps").reshape(-1), interp.philox_normal(77, step, B * z, rank=1))
        assert not np.array_equal(interp.philox_normal(77, step, B * z, rank=1), interp.philox_normal(77, step, B * z, rank=0))      # ranks draw independent noise
        assert int(H.get_io(plan, A, "cursor")[0]) == step + 1
    with pytest.raises(ValueError):
        planner.lower(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(resident_units=3))


def test_philox_known_answers_and_moments():
    """Philox4x32-10 against the published Random123 known-answer vectors; the derived normals have the right moments."""
    z, f = np.zeros(1), np.full(1, 0xFFFFFFFF)
    assert [int(v[0]) for v in interp.philox4x32_10(z, z, z, z, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(v[0]) for v in interp.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    pi = [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]
    got = interp.philox4x32_10(*[np.full(1, v) for v in pi], 0xA4093822, 0x299F31D0)
    assert [int(v[0]) for v in got] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    x = interp.philox_normal(1234, 7, 1 << 20).astype(np.float64)
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1) < 5e-3 and abs((x ** 3).mean()) < 1e-2 and abs((x ** 4).mean() - 3) < 3e-2
    assert not np.array_equal(interp.philox_normal(1234, 8, 64), interp.philox_normal(1234, 7, 64))


def test_multimodal_towers_are_zipped_into_pairs(tmp_path):
    """planner.Builder.zip_towers: the two towers' same-kind launches become HP_OP_PAIR units (fewer launches, same records otherwise), the
    zipped program validates in the library, and the ordered-slab weight gradients (one slab shared by both towers) stay unzipped."""
    import ctypes
    from hippie_amd import export
    cfg = planner.ModelCfg("multimodal", 10, 50, 100)

    def launches(plan):
        ops = plan.ops.array()
        return sum(1 for r in ops if not int(r["flags"]) & P.FLAG_MEMBER), sum(1 for r in ops if int(r["op"]) == P.PAIR), ops

    on = planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0))
    off = planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0, zip_towers=False))
    (l_on, p_on, ops_on), (l_off, p_off, ops_off) = launches(on), launches(off)
    assert l_on < 0.75 * l_off and p_on > p_off + 60
    # the same work: every non-PAIR record of the unzipped program is in the zipped one (flags aside: members are marked)
    key = lambda r: (int(r["op"]), tuple(int(v) for v in r["i"]), tuple(int(b) for b in r["buf"]))      # noqa: E731
    non_pair = lambda ops: sorted(key(r) for r in ops if int(r["op"]) not in (P.PAIR, P.WGRAD_GROUP))     # noqa: E731  (those two hold record indices)
    # (workspace offsets differ after liveness packing, so compare with packing off)
    a = planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0, reuse_workspace=False)).ops.array()
    b = planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0, reuse_workspace=False, zip_towers=False)).ops.array()
    assert non_pair(a) == non_pair(b)
    for k, r in enumerate(ops_on):
        if int(r["op"]) == P.PAIR:
            i, j = int(r["i"][0]), int(r["i"][1])
            assert i < k and j < k and i != j and int(ops_on[i]["op"]) == int(ops_on[j]["op"])
            assert int(ops_on[i]["flags"]) & P.FLAG_MEMBER and int(ops_on[j]["flags"]) & P.FLAG_MEMBER
    path = str(tmp_path / "mm.hpm")
    export.save_model(on, path)
    lib = P.load_library()
    m = ctypes.c_void_p()
    assert lib.hp_model_load(path.encode(), export.NO_DEVICE, ctypes.byref(m)) == 0, lib.hp_last_error()
    lib.hp_model_destroy(m)
    det_on = launches(planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0, deterministic_wgrad=True)))
    det_off = launches(planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0, deterministic_wgrad=True, zip_towers=False)))
    bwd = lambda plan: plan.ops.segments["bwd"]                                                            # noqa: E731
    assert det_on[1] > det_off[1]            # forward towers are still zipped ...
    ops = det_on[2]
    f, c = bwd(planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, clip=1.0, deterministic_wgrad=True)))
    slab_users = [k for k in range(f, f + c) if int(ops[k]["op"]) in (P.WGRAD_TAPS, P.SLAB_REDUCE)]
    assert slab_users and all(int(ops[slab_users[q + 1]]["op"]) == P.SLAB_REDUCE for q in range(0, len(slab_users) - 1, 2) if int(ops[slab_users[q]]["op"]) == P.WGRAD_TAPS)
