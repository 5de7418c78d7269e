"""The mask-injected oracle (oracle/cvae_oracle.py: Ctx.lrelu / _MaskedLeakyReLU) that the GPU gradient-parity
tests rely on, checked on CPU: injecting a model's own leaky-ReLU branches changes nothing; the branches read
back from an executed program (here: the numpy interpreter's arenas, on the GPU: the engine's workspace) cover
every site of the oracle; a forced flip of one near-zero activation moves upstream gradients while leaving the
loss where it was — which is exactly why unmasked gradient comparisons are undecidable."""
import numpy as np
import pytest
import torch

from hippie_amd import planner, program as P
from oracle import cvae_oracle as O
from oracle import interp
from tests import helpers as H

torch.set_num_threads(4)


def _setup(kind, z, L, B, L2=None, salt=1):
    cfg = planner.ModelCfg(kind=kind, z_dim=z, output_size=L, output_size2=L2 or 100)
    plan = planner.lower(cfg, B, planner.TrainCfg())
    ops = plan.ops.array()
    A = H.make_arenas(plan)
    om = O.OracleModel(kind, z, L, output_size2=L2, salt=salt, dtype=torch.float64)
    H.load_state(plan, A, om.state)
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt, name="x" if kind == "unimodal" else "x1")
    for name, v in (("x", x), ("src", src), ("cls", cls), ("eps", eps)):
        H.set_io(plan, A, name, v.numpy())
    batch = (x.double(), src, None)
    if kind != "unimodal":
        x2 = O.synth_inputs(B, L2, z, salt=salt, name="x2")[0]
        H.set_io(plan, A, "x2", x2.numpy())
        batch = (x.double(), x2.double(), src, None)
    s, c = plan.ops.segments["fwd_train"]
    interp.run(ops, A, s, c)
    return plan, ops, A, om, batch, eps.double()


@pytest.mark.parametrize("kind,z,L,B,L2", [("unimodal", 10, 50, 8, None), ("multimodal", 10, 50, 6, 100)])
def test_own_branches_change_nothing_and_cover_every_site(kind, z, L, B, L2):
    plan, ops, A, om, batch, eps = _setup(kind, z, L, B, L2)
    masks = H.arena_masks(plan, ops, A)
    taps = {}
    outs = om.forward(batch, eps, True, taps=taps)
    om.losses(batch, outs)[0].backward()
    g0 = {k: v.clone() for k, v in om.grads().items() if v is not None}
    assert H.count_mask_flips(masks, taps)[0] == 0
    om2 = O.OracleModel(kind, z, L, output_size2=L2, salt=1, dtype=torch.float64)
    ctx_outs = om2.forward(batch, eps, True, masks=masks)       # KeyError if a site had no mask
    om2.losses(batch, ctx_outs)[0].backward()
    for k, g in g0.items():
        assert torch.equal(om2.grads()[k], g), k
    # every site the oracle visited has a mask of the activation's shape, and nothing else is in the dict
    ctx = O.Ctx(True, None, masks)
    if kind == "unimodal":
        O.cvae_forward(om2.state, batch[0], batch[1], None, eps, ctx)
    else:
        O.mm_forward(om2.state, batch[0], batch[1], batch[2], None, eps, ctx)
    assert sorted(ctx.sites) == sorted(masks)


def test_one_forced_flip_moves_gradients_but_not_the_loss():
    plan, ops, A, om, batch, eps = _setup("unimodal", 10, 50, 8)
    masks = H.arena_masks(plan, ops, A)
    taps = {}
    outs = om.forward(batch, eps, True, taps=taps)
    loss0 = om.losses(batch, outs)[0]
    loss0.backward()
    g0 = {k: v.clone() for k, v in om.grads().items() if v is not None}
    site = "decoder.layer3.0.bn2"
    act = taps[site].detach()
    idx = np.unravel_index(int(act.abs().argmin()), act.shape)      # the activation closest to zero
    flipped = {k: m.clone() for k, m in masks.items()}
    flipped[site][idx] = ~flipped[site][idx]
    om2 = O.OracleModel("unimodal", 10, 50, salt=1, dtype=torch.float64)
    outs2 = om2.forward(batch, eps, True, masks=flipped)
    loss2 = om2.losses(batch, outs2)[0]
    loss2.backward()
    assert abs(float(loss2) - float(loss0)) <= 1e-3 * abs(float(loss0))
    moved = max(float((om2.grads()[k] - g).abs().max() / g.abs().max().clamp_min(1e-30)) for k, g in g0.items()
                if k.startswith(("encoder.", "decoder.layer4")))
    assert moved > 1e-6, "a flipped branch must change upstream gradients"


def test_wrong_mask_shape_is_rejected():
    plan, ops, A, om, batch, eps = _setup("unimodal", 10, 50, 8)
    masks = H.arena_masks(plan, ops, A)
    masks["encoder.bn1"] = masks["encoder.bn1"][:, :, :-1]
    with pytest.raises(AssertionError):
        om.forward(batch, eps, True, masks=masks)


def test_flip_budget_passes_for_the_interpreter_and_catches_a_mis_signed_band():
    """helpers.assert_flip_budget — the unmasked anchor of the GPU gradient tests — on the numpy interpreter's forward:
    no sign differs from the free-running float64 oracle; an implementation that took the wrong branch on a band of
    clearly non-zero inputs (which the mask-injected oracle would FOLLOW) fails it, and so does one that exceeds the
    count budget on values below the parity bar."""
    plan, ops, A, om, batch, eps = _setup("unimodal", 10, 50, 8, None)
    pres = H.pre_activations(plan, ops, lambda off, n: A.mem[P.WS][off: off + 4 * n].view(np.float32), plan.B)
    taps = []
    for dt in (torch.float32, torch.float64):
        o = O.OracleModel("unimodal", 10, 50, salt=1, dtype=dt)
        taps.append({})
        b = tuple(t.to(dt) if (t is not None and t.is_floating_point()) else t for t in batch)
        with torch.no_grad():
            o.forward(b, eps.to(dt), True, taps=taps[-1])
    flips, elems, _ = H.assert_flip_budget(pres, taps[0], taps[1], "interp")
    assert flips == 0 and elems == sum(v.size for v in pres.values())
    key = "encoder.layer2.0.bn1"
    bad = {k: v.copy() for k, v in pres.items()}
    big = np.abs(bad[key]) > 0.5 * np.abs(bad[key]).max()
    bad[key][big] *= -1.0                                     # a mis-signed band of clearly non-zero activations
    with pytest.raises(AssertionError):
        H.assert_flip_budget(bad, taps[0], taps[1], "mis-signed")
    tiny = {k: v.copy() for k, v in pres.items()}
    small = np.abs(tiny[key]) < 1e-3 * np.abs(tiny[key]).max()
    assert small.sum() > H.flip_allowance(elems)
    tiny[key][small] *= -1.0                                  # many flips of small values: over the count budget
    with pytest.raises(AssertionError):
        H.assert_flip_budget(tiny, taps[0], taps[1], "too many")
