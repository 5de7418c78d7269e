"""BalancedBatchSampler (hippie/dataloading.py:107-151): index streams identical to the reference's own
(tests/golden/balanced_sampler.npz, generator make_golden_sampler.py) for the same `random` seed."""
import os
import random

import numpy as np
import pytest
import torch

from hippie_amd.dataloading import BalancedBatchSampler

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "balanced_sampler.npz"))
CASES = ["skewed", "two_class", "already_balanced", "single_class"]


@pytest.mark.parametrize("name", CASES)
def test_index_stream_matches_reference(name):
    labels = torch.as_tensor(G[name + "_labels"]).long()
    random.seed(int(G[name + "_seed"]))
    s = BalancedBatchSampler(list(range(len(labels))), labels)
    assert len(s) == int(G[name + "_len"])
    assert s.keys == G[name + "_keys"].tolist()
    np.testing.assert_array_equal(np.array(list(s)), G[name + "_epoch1"])
    np.testing.assert_array_equal(np.array(list(s)), G[name + "_epoch2"])


def test_balance_property_and_errors():
    rng = np.random.default_rng(3)
    labels = torch.as_tensor(rng.choice(6, size=10_000, p=[0.4, 0.3, 0.15, 0.1, 0.04, 0.01])).long()
    s = BalancedBatchSampler(range(len(labels)), labels)
    idx = np.array(list(s))
    counts = np.bincount(labels.numpy()[idx], minlength=6)
    assert len(idx) == len(s) and (counts == counts[0]).all() and counts[0] == s.balanced_max
    # every original sample of the largest class appears exactly once; round-robin interleave
    big = int(np.bincount(labels.numpy()).argmax())
    assert sorted(idx[labels.numpy()[idx] == big].tolist()) == np.nonzero(labels.numpy() == big)[0].tolist()
    assert (labels.numpy()[idx[: 6]] == np.array(s.keys)).all()
    with pytest.raises(Exception, match="pass the tensor of labels"):
        BalancedBatchSampler(range(4))
    assert list(BalancedBatchSampler([], torch.zeros(0).long())) == []
