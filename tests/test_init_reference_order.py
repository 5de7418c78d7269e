"""Constructor initialisation in the reference's RNG order (`torch.manual_seed(42)` -> random_split -> wave
model -> time model, scripts/train_model_with_multimodal.py:78,136-176): hippie_amd.model.reference_init_state against
checksums of the REAL reference classes (tests/golden/init_seed42.npz, generator: tests/golden/make_golden_init.py)."""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import random_split

from hippie_amd import planner
from hippie_amd.model import reference_init_state, reference_param_order

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "init_seed42.npz")))


def checks(t):
    f = t.detach().double().reshape(-1)
    return np.array([float(f.sum()), float(f.abs().sum()), float(f[0]), float(f[min(1, len(f) - 1)]), float(f[min(2, len(f) - 1)]), float(f[-1])])


def assert_same(got, ref):
    """sampled elements (first three, last) bit for bit; the float64 sums up to the summation order of torch's
    multi-threaded reduction"""
    np.testing.assert_array_equal(got[:, 2:], ref[:, 2:])
    np.testing.assert_allclose(got[:, :2], ref[:, :2], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n_pool,z", [(280, 5), (3797, 10), (15631, 10)])
def test_seed42_split_then_wave_then_time_init_is_the_references(n_pool, z):
    tag = f"pool{n_pool}_z{z}"
    torch.manual_seed(42)
    n_tr = int(0.8 * n_pool)
    tr, te = random_split(list(range(n_pool)), [n_tr, n_pool - n_tr])
    np.testing.assert_array_equal(tr.indices[:16], G[tag + ".train_idx_head"])
    for name, L in (("wave", 50), ("time", 100)):
        cfg = planner.ModelCfg("unimodal", z, L, 0, 5, 5, 5)
        sd = reference_init_state(cfg)
        assert list(sd) == [str(k) for k in G[f"{tag}.{name}.names"]]          # construction order == state_dict order
        assert_same(np.stack([checks(v) for v in sd.values()]), G[f"{tag}.{name}.checks"])
    # and exactly as many draws were consumed as the reference's constructors consume
    assert int(torch.empty((), dtype=torch.int64).random_().item()) == int(G[tag + ".next_draw"])


def test_multimodal_init_is_the_references():
    torch.manual_seed(42)
    cfg = planner.ModelCfg("multimodal", 10, 50, 100, 5, 5, 5)
    sd = reference_init_state(cfg)
    assert list(sd) == [str(k) for k in G["multimodal_z10.names"]]
    assert_same(np.stack([checks(v) for v in sd.values()]), G["multimodal_z10.checks"])
    assert int(torch.empty((), dtype=torch.int64).random_().item()) == int(G["multimodal_z10.next_draw"])


def test_order_covers_exactly_the_planner_parameters():
    for cfg in (planner.ModelCfg("unimodal", 10, 50), planner.ModelCfg("multimodal", 32, 256, 32)):
        plan = planner.lower(cfg, 4)
        order = reference_param_order(cfg)
        assert {k for k, _, _ in order} == set(plan.params)
        for k, shape, _ in order:
            assert tuple(plan.params[k].shape) == tuple(shape), k
