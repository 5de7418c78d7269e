"""GPU: hippie_amd.streams — measuring which pair of HIP streams overlaps must not touch any model state, and the streams it
returns must give the same numbers as the default stream."""
import numpy as np
import pytest
import torch

from hippie_amd import planner, streams
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O

pytestmark = pytest.mark.gpu


def engines():
    out = []
    for L, salt in ((50, 1), (100, 2)):
        e = Engine(planner.ModelCfg(kind="unimodal", z_dim=10, output_size=L), 32, planner.TrainCfg(lr=1e-4, clip=1.0))
        om = O.OracleModel("unimodal", 10, L, salt=salt)
        e.load_state_dict({k: v.detach() for k, v in om.state.items()})
        x, src, _, eps = O.synth_inputs(32, L, 10, salt=salt)
        e.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
        out.append(e)
    return out


def snapshot(e):
    torch.cuda.synchronize()
    return [t.detach().cpu().numpy().copy() for t in (e.params, e.bufs, e.m, e.v, e.grads)] + [dict(e.num_batches_tracked)]


def test_pick_leaves_state_alone_and_reports():
    engs = engines()
    for e in engs:
        e.train_step(True)                  # gradients, moments and running statistics are non-trivial
    before = [snapshot(e) for e in engs]
    streams._CHOSEN.clear()
    rep = {}
    ss = streams.pick_concurrent_streams(engs, report=rep)
    assert len(ss) == 2 and ss[0] != ss[1] and all(isinstance(s, torch.cuda.Stream) for s in ss)
    assert rep["us"] > 0 and rep["serial_us"] > 0 and f"{rep['chosen'][0]},{rep['chosen'][1]}" in rep["tried"]
    assert rep["us"] == min(rep["tried"].values()) or rep["us"] <= 0.85 * rep["serial_us"]
    for e, b in zip(engs, before):
        for x, y in zip(snapshot(e), b):
            if isinstance(x, dict):
                assert x == y
            else:
                np.testing.assert_array_equal(x, y)
    # cached for the life of the process; refresh measures again
    rep2 = {}
    ss2 = streams.pick_concurrent_streams(engs, report=rep2)
    assert ss2 == ss and rep2.get("cached") is True
    assert len(streams.pick_concurrent_streams(engs[:1])) == 1
    rep3 = {}
    streams.pick_concurrent_streams(engs, report=rep3, refresh=True)
    assert "cached" not in rep3


def test_steps_on_the_picked_streams_equal_steps_on_the_default_stream():
    """one optimisation step per model, side by side on the picked streams, against the same step on the default stream: losses and
    gradients agree to the run-to-run level of the fp32 atomics in the grouped weight gradients (not bit for bit: their order is free)"""
    a, b = engines(), engines()
    for e in a:
        e.train_step(True)
    ss = streams.pick_concurrent_streams(b)
    cur = torch.cuda.current_stream()
    for s in ss:
        s.wait_stream(cur)
    for e, s in zip(b, ss):
        with torch.cuda.stream(s):
            e.train_step(True)
    for s in ss:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        np.testing.assert_allclose(y.scalars(), x.scalars(), rtol=1e-6)
        gx, gy = x.grads.cpu().numpy(), y.grads.cpu().numpy()
        assert np.abs(gx - gy).max() <= 1e-5 * np.abs(gx).max()
        assert x.num_batches_tracked == y.num_batches_tracked
