"""CPU: the zipped wave+time program (hippie_amd.pair) equals the two separate programs when run through
the numpy interpreter, and validates in the C library."""
import ctypes

import numpy as np
import torch

from hippie_amd import pair, planner, program as P
from oracle import cvae_oracle as O
from oracle import interp
from tests import helpers as H


def test_zipped_program_equals_separate_programs():
    B, z = 4, 10
    tcs = [planner.TrainCfg(lr=1e-3, intra_pair=False, chain_small=False, group_small_wgrads=False, fuse_heads=False),
           planner.TrainCfg(lr=1e-3, clip=1.0, intra_pair=False, chain_small=False, group_small_wgrads=False, fuse_heads=False)]      # (as PairEngine lowers them)
    plans = [planner.lower(planner.ModelCfg("unimodal", z, L), B, tc) for L, tc in zip((50, 100), tcs)]
    ops, segments, notes, bases_b = pair.zip_programs(*plans)
    assert len(notes) == len(ops)
    n_pair = sum(int(r["op"]) == P.PAIR for r in ops)
    n_group = sum(int(r["op"]) == P.WGRAD_GROUP for r in ops)
    assert n_pair > 100 and n_group == 2
    sa, sb = pair.arena_sizes(plans[0]), pair.arena_sizes(plans[1])
    # the C library accepts it (validation needs no GPU)
    lib = P.load_library()
    bases = (ctypes.c_void_p * 6)(*[ctypes.c_void_p(0x1000)] * 6)
    sz = (ctypes.c_int64 * 6)(*[a + b for a, b in zip(sa, sb)])
    h = ctypes.c_void_p()
    rc = lib.hp_program_create(ops.ctypes.data_as(ctypes.c_void_p), len(ops), bases, sz, ctypes.byref(h))
    assert rc == 0, lib.hp_last_error().decode()
    lib.hp_program_destroy(h)
    # joint run in the interpreter vs separate runs
    J = interp.Arenas([a + b for a, b in zip(sa, sb)])
    singles = []
    for k, (plan, L) in enumerate(zip(plans, (50, 100))):
        A = H.make_arenas(plan)
        om = O.OracleModel("unimodal", z, L, salt=k)
        H.load_state(plan, A, om.state)
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=k)
        for nme, v in (("x", x), ("src", src), ("cls", cls), ("eps", eps)):
            H.set_io(plan, A, nme, v.numpy())
        base = [0] * 6 if k == 0 else bases_b
        for sp in range(6):
            J.mem[sp][base[sp]: base[sp] + A.mem[sp].size] = A.mem[sp]
        for seg in ("fwd_train", "bwd", "opt"):
            s, c = plan.ops.segments[seg]
            interp.run(plan.ops.array(), A, s, c)
        singles.append(A)
    for seg in ("fwd_train", "bwd", "opt"):
        s, c = segments[seg]
        interp.run(ops, J, s, c)
    for k, A in enumerate(singles):
        base = [0] * 6 if k == 0 else bases_b
        for sp in (P.PARAM, P.GRAD, P.BUF, P.ADAM_M, P.ADAM_V):
            np.testing.assert_array_equal(J.mem[sp][base[sp]: base[sp] + A.mem[sp].size], A.mem[sp], err_msg=f"model {k} space {sp}")
