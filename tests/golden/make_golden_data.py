"""Golden vectors for the preprocessing row (f1): the reference's own EphysDatasetLabeled
(hippie/dataloading.py:62-104) applied to the first rows of every shipped dataset, read exactly as the
scripts read them (pd.read_csv WITHOUT index_col, scripts/train_model_with_multimodal.py:117-121 — so the
unnamed index column of some CSVs becomes feature 0).  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_data.py"""
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from hippie.dataloading import EphysDatasetLabeled      # noqa: E402  (the reference)

ROOT = "/root/reference/datasets"
out = {}
names = []
for name in sorted(os.listdir(ROOT)):
    wf, isi = os.path.join(ROOT, name, "waveforms.csv"), os.path.join(ROOT, name, "isi_dist.csv")
    if not (os.path.exists(wf) and os.path.exists(isi)):
        continue
    w = pd.read_csv(wf).to_numpy()[:8]
    t = pd.read_csv(isi).to_numpy()[:8]
    lab = np.arange(len(w))
    ds_w = EphysDatasetLabeled(w, t, lab, mode="wave", normalize=False)
    ds_t = EphysDatasetLabeled(w, t, lab, mode="time", normalize=False)
    out[name + ".wave_in"] = w.astype(np.float64)
    out[name + ".isi_in"] = t.astype(np.float64)
    out[name + ".wave_out"] = torch.stack([ds_w[i][0] for i in range(len(w))]).numpy()
    out[name + ".isi_out"] = torch.stack([ds_t[i][0] for i in range(len(w))]).numpy()
    names.append(name)
    print(name, w.shape, t.shape, out[name + ".wave_out"].shape, out[name + ".isi_out"].shape)
out["names"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "datasets_first8.npz"), **out)

# (9) the reference's train/val split: torch.manual_seed(42) then random_split of the index list
# (scripts/train_model_with_multimodal.py:78,136-147) for the pool sizes the shipped datasets give
from torch.utils.data import random_split   # noqa: E402
splits = {}
for n, prop in ((2975, 0.8), (3797, 0.8), (392, 0.1)):
    torch.manual_seed(42)
    idx = list(range(n))
    tr, te = random_split(idx, [int(prop * n), n - int(prop * n)])
    splits[f"n{n}_train"] = np.array(tr.indices)
    splits[f"n{n}_test"] = np.array(te.indices)
np.savez_compressed(os.path.join(HERE, "random_split_seed42.npz"), **splits)
print("splits", {k: v.shape for k, v in splits.items()})
