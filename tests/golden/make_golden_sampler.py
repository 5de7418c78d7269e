"""Golden index streams of the reference's BalancedBatchSampler (hippie/dataloading.py:107-151).

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_sampler.py
"""
import os
import random
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from hippie.dataloading import BalancedBatchSampler  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    rng = np.random.default_rng(7)
    cases = {
        "skewed": rng.choice(5, size=97, p=[0.5, 0.25, 0.15, 0.07, 0.03]),
        "two_class": np.array([1] * 9 + [0] * 2 + [1] * 4),
        "already_balanced": np.tile(np.arange(4), 6),
        "single_class": np.zeros(5, dtype=np.int64),
    }
    for name, labels in cases.items():
        lab = torch.as_tensor(labels).long()
        random.seed(1000 + len(name))
        s = BalancedBatchSampler(list(range(len(lab))), lab)
        first = list(s)
        second = list(s)              # a second epoch over the same object
        out[name + "_seed"] = np.int64(1000 + len(name))
        out[name + "_labels"] = labels.astype(np.int64)
        out[name + "_epoch1"] = np.array(first, dtype=np.int64)
        out[name + "_epoch2"] = np.array(second, dtype=np.int64)
        out[name + "_len"] = np.int64(len(s))
        out[name + "_keys"] = np.array(s.keys, dtype=np.int64)
        print(name, len(s), first[:10])
    np.savez_compressed(os.path.join(HERE, "balanced_sampler.npz"), **out)


if __name__ == "__main__":
    main()
