"""The shipped tables of BASELINE configs[0] / [1]'s target dataset as a fixture (data, not code): datasets/cellexplorer-celltype/
{waveforms,isi_dist}.csv — 392 units, read exactly as the fine-tune stage reads them (pd.read_csv WITHOUT index_col + dropna(axis=1),
scripts/train_model_with_multimodal.py:234-236: the unnamed index column is feature 0) — together with what the reference's OWN
EphysDatasetLabeled (hippie/dataloading.py:62-104) yields for every row.  tests/test_gpu_real_data.py runs preprocessing, the label-free
fine-tune split (39 / 353, random_split_seed42.npz) and the embedding pass on them.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_cellexplorer.py"""
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from hippie.dataloading import EphysDatasetLabeled      # noqa: E402  (the reference)

root = "/root/reference/datasets/cellexplorer-celltype"
w = pd.read_csv(os.path.join(root, "waveforms.csv")).dropna(axis=1).to_numpy()
t = pd.read_csv(os.path.join(root, "isi_dist.csv")).dropna(axis=1).to_numpy()
lab = np.zeros(len(w))
ds_w = EphysDatasetLabeled(w, t, lab, mode="wave", normalize=False)
ds_t = EphysDatasetLabeled(w, t, lab, mode="time", normalize=False)
out = {"wave_in": w.astype(np.float64), "isi_in": t.astype(np.float64),
       "wave_out": torch.stack([ds_w[i][0] for i in range(len(w))]).numpy(), "isi_out": torch.stack([ds_t[i][0] for i in range(len(w))]).numpy()}
assert out["wave_in"].shape == (392, 47) and out["isi_in"].shape == (392, 100), (out["wave_in"].shape, out["isi_in"].shape)
np.savez_compressed(os.path.join(HERE, "cellexplorer_celltype_tables.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "cellexplorer_celltype_tables.npz")), "bytes")
print("wave range", out["wave_out"].min(), out["wave_out"].max(), "isi range", out["isi_out"].min(), out["isi_out"].max())
