"""planner.pack_workspace (liveness-based reuse of the workspace arena): a packed program computes bit-for-bit what the
unpacked one computes, never reads workspace it has not written (NaN-poisoned arena), shares memory only between
allocations whose live ranges are disjoint, and is smaller."""
import dataclasses

import numpy as np
import pytest

from hippie_amd import planner, program as P
from oracle import cvae_oracle as O
from oracle import interp
from tests import helpers as H

MASK = (1 << 56) - 1
CASES = {
    "wave": (planner.ModelCfg(kind="unimodal", z_dim=10, output_size=50), 6, dict(clip=0.0)),
    "time_clip_unfused": (planner.ModelCfg(kind="unimodal", z_dim=5, output_size=100), 5, dict(clip=1.0, fuse_bn=False)),
    "multi": (planner.ModelCfg(kind="multimodal", z_dim=10, output_size=50, output_size2=100), 4, dict(clip=1.0)),
    "wave_slabs": (planner.ModelCfg(kind="unimodal", z_dim=10, output_size=50), 6, dict(deterministic_wgrad=True)),
}


def run_program(cfg, B, tc, poison):
    plan = planner.lower(cfg, B, tc)
    ops = plan.ops.array()
    A = H.make_arenas(plan)
    if poison:
        A.mem[P.WS][:] = 0xFF               # every float / double of the workspace reads as NaN until an op writes it
    om = O.OracleModel(cfg.kind, cfg.z_dim, cfg.output_size, output_size2=cfg.output_size2 if cfg.kind == "multimodal" else None, salt=3)
    H.load_state(plan, A, om.state)
    x, src, cls, eps = O.synth_inputs(B, cfg.output_size, cfg.z_dim, salt=3, name="x1")
    H.set_io(plan, A, "x", x.numpy())
    if cfg.kind == "multimodal":
        H.set_io(plan, A, "x2", O.synth_inputs(B, cfg.output_size2, cfg.z_dim, salt=3, name="x2")[0].numpy())
    H.set_io(plan, A, "src", src.numpy())
    H.set_io(plan, A, "cls", cls.numpy())
    H.set_io(plan, A, "eps", eps.numpy())
    out = {}
    # eval between the training forward and its backward: the two passes must not share memory
    for seg in ("fwd_train", "fwd_eval", "bwd", "opt", "fwd_eval"):
        s, c = plan.ops.segments[seg]
        interp.run(ops, A, s, c)
        if seg == "fwd_train":
            out["rec_train"] = H.get_io(plan, A, "rec_train")
            out["scalars"] = H.get_io(plan, A, "scalars")
    out["rec_eval"] = H.get_io(plan, A, "rec_eval")
    out["enc_eval"] = H.get_io(plan, A, "enc_eval")
    out["grads"] = A.mem[P.GRAD].view(np.float32).copy()
    out["params"] = A.mem[P.PARAM].view(np.float32).copy()
    out["bufs"] = A.mem[P.BUF].view(np.float32).copy()
    return plan, out


@pytest.mark.parametrize("name", list(CASES))
def test_packed_program_equals_unpacked_and_reads_only_what_it_wrote(name):
    cfg, B, kw = CASES[name]
    tc = planner.TrainCfg(lr=1e-3, **kw)
    plan_u, ref = run_program(cfg, B, dataclasses.replace(tc, reuse_workspace=False), poison=False)
    plan_p, got = run_program(cfg, B, tc, poison=True)
    assert plan_u.ws_unpacked is None and plan_p.ws_unpacked == plan_u.ws_bytes
    # the statistics region (32 MB whatever the batch) and the weight-fragment images (12 bytes per conv weight whatever the batch: written at
    # the head of every forward pass, read to the end of the backward pass) dominate at test sizes
    seen = {}
    for r in plan_u.ops.recs:
        if int(r["op"]) == P.WFRAG:
            T_, N_, K_, which = [int(v) for v in r["i"][:4]]
            if which & 1:
                seen[int(r["buf"][1])] = T_ * (K_ // 16) * (-(-N_ // 32)) * 3072
            if which & 2:
                seen[int(r["buf"][2])] = T_ * ((-(-N_ // 32)) * 2) * (K_ // 32) * 3072
    fixed = plan_u.stats_cap + sum(seen.values())
    assert plan_p.ws_bytes - fixed < 0.8 * (plan_u.ws_bytes - fixed), (plan_p.ws_bytes, plan_u.ws_bytes)
    for k, v in ref.items():
        assert np.isfinite(got[k]).all(), f"{k}: an op read workspace nothing had written"
        np.testing.assert_array_equal(got[k], v, err_msg=k)


def test_shared_memory_only_between_disjoint_live_ranges():
    """Independent restatement of the invariant.  The unpacked lowering gives every operand an allocation of its own
    (identity); the packed lowering of the same model gives the memory it ends up in.  Two allocations whose packed
    memory overlaps must have record ranges that do not intersect (counted where the records execute)."""
    cfg, B, kw = CASES["wave"]
    tc = planner.TrainCfg(lr=1e-3, **kw)
    unpacked = planner.lower(cfg, B, dataclasses.replace(tc, reuse_workspace=False))
    packed = planner.lower(cfg, B, tc)
    ru, rp = unpacked.ops.recs, packed.ops.recs
    assert len(ru) == len(rp)
    ident = sorted((a[0], a[0] + a[1]) for a in unpacked.allocs)
    at = list(range(len(ru)))
    for g, r in enumerate(ru):
        if int(r["op"]) in (P.WGRAD_GROUP, P.HEADS):
            for k in range(int(r["i"][0]), int(r["i"][0]) + int(r["i"][1])):
                at[k] = g
        elif int(r["op"]) == P.PAIR:
            at[int(r["i"][0])] = at[int(r["i"][1])] = g
        ngroup = (int(r["flags"]) >> P.FLAG_GROUP_SHIFT) & P.FLAG_GROUP_MASK      # small-leaf group: members run at its last record
        for k in range(g - ngroup, g):
            at[k] = g
    assert any((int(r["flags"]) >> P.FLAG_GROUP_SHIFT) & P.FLAG_GROUP_MASK for r in ru), "the default lowering groups its small leaves"
    span, mem = {}, {}
    for k, (u, p) in enumerate(zip(ru, rp)):
        for bu, bp in zip(u["buf"], p["buf"]):
            bu, bp = int(bu), int(bp)
            if bu == P.NULL or (bu >> 56) != P.WS:
                assert bu == bp
                continue
            off = bu & MASK
            (a,) = [a for a in ident if a[0] <= off < a[1]]
            lo, hi = span.get(a, (at[k], at[k]))
            span[a] = (min(lo, at[k]), max(hi, at[k]))
            base = (bp & MASK) - (off - a[0])
            assert mem.setdefault(a, base) == base, "one allocation, two places"
    items = sorted((mem[a], mem[a] + a[1] - a[0], span[a]) for a in span)
    shared = 0
    for i, (lo, hi, sa) in enumerate(items):
        for lo2, hi2, sb in items[i + 1:]:
            if lo2 >= hi:
                break
            shared += 1
            assert sa[1] < sb[0] or sb[1] < sa[0], f"[{lo},{hi}) records {sa} and [{lo2},{hi2}) records {sb} overlap in memory and in time"
    assert shared > 0, "nothing is shared: the pass did not run"
